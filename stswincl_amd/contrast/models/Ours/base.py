"""Drop-in for pixcontrast_18/contrast/models/Ours/base.py: ``TswinPlusv5`` is TswinPlus with the contrastive
package's default feature resolution (32, 56) (swin_tem.py:281)."""
from ....net.Ours.base18 import TswinPlus, decode_tokens  # noqa: F401


class TswinPlusv5(TswinPlus):
    def __init__(self, num_classes, input_resolution=(32, 56)):
        super().__init__(num_classes, input_resolution)
