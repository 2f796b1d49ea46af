import math


def mse2psnr(mse):
    """utils/metrics.py:11-15 of the reference."""
    if mse == 0:
        mse = 1e-5
    return -10.0 * math.log10(mse)
