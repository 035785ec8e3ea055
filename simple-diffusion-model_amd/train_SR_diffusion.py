"""Train a super-resolution (cold-diffusion residual) U-Net on MI355X -- same command line and JSON keys as the
reference's train_SR_diffusion.py (extra keys ``lr_dim``, ``sr_dim``, ``cond_t``; :60-62).

Step body (reference :318-374): lowres = area(area(x0, lr_dim), sr_dim); the 6-channel input is
cat(q(x0, t, eps), q(lowres, cond_t, eps)); the Tanh-headed net regresses x0 - lowres with MSE.  Area
interpolation, both q-samples, forward, backward and Adam run as libsdm_hip kernels (sdm_hip.trainer with
``sr=(lr_dim, sr_dim, cond_t)``); sample grids come from the fused cold-diffusion sampler.
"""
from sdm_hip.cli_train import run_training


def main():
    run_training(sr=True)


if __name__ == "__main__":
    main()
