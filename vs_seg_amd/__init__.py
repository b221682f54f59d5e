"""vs_seg_amd — MI355X-native (gfx950) hot path of KCL-BMEIS/VS_Seg behind the reference's own Python interfaces.

    from vs_seg_amd import UNet2d5_spvPA, Dice_spvPA, sliding_window_inference, Adam

`UNet2d5_spvPA` / `Dice_spvPA` mirror ref:params/networks/nets/unet2d5_spvPA.py and ref:params/losses/dice_spvPA.py;
`sliding_window_inference` mirrors the MONAI 0.4.0 function called at ref:params/VSparams.py:568-574.  All compute is
in `libvsseg_hip.so` (hand-written HIP, C ABI in include/vsseg_hip.h); there is no CPU fallback.
"""
from ._lib import VssegError, fx_status  # noqa: F401
from .inferers import compute_dice_score, sliding_window_inference  # noqa: F401
from .losses.dice_spvPA import Dice_spvPA  # noqa: F401
from .networks.nets.unet2d5_spvPA import UNet2d5_spvPA  # noqa: F401
from .optim import Adam  # noqa: F401

__all__ = ["UNet2d5_spvPA", "Dice_spvPA", "sliding_window_inference", "compute_dice_score", "Adam", "fx_status", "VssegError"]
