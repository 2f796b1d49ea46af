from .cfgnode import CfgNode
from .metrics import mse2psnr
from .tensorf_utils import TVLoss, N_to_reso
