import faulthandler, os, socket, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import torch.multiprocessing as mp


def worker(rank, world, port, q):
    faulthandler.dump_traceback_later(60, exit=True)
    import test_hip_contrast_bank as T
    T._world_token_worker(rank, world, port, q)
    print("rank", rank, "done", flush=True)


if __name__ == "__main__":
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs: p.start()
    for p in procs: p.join(120)
    while not q.empty(): print(q.get())
