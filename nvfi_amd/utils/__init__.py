from .cfgnode import CfgNode
from .metrics import mse2psnr
from .tensorf_utils import TVLoss, N_to_reso, mse_loss
from .evaluation_utils import save_checkpoint, load_checkpoint, load_model_checkpoint, render_test_evaluation
from .segm_utils import sample_volume_points, balanced_sample, segm_points
