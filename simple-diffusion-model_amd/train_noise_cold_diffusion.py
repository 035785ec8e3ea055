"""Train an x0-prediction ("cold diffusion") U-Net on MI355X — same command line and JSON config keys as the reference's
train_noise_cold_diffusion.py (no ``diffusion_alg`` key; checkpoints plot through ``cold_diffusion_sampling``).

The step body (train_noise_cold_diffusion.py:307-348: x_t = q(x_t | x_0), x0_hat = U_Net(x_t, t, labels), MSE against the
clean image) is sdm_hip.trainer.DiffusionTrainer with ``objective="x0"``: the same kernels as noise-prediction training,
a different regression target.  Data parallel under ``torchrun`` and ``synthetic:N:C:H:W`` datasets as in
train_diffusion.py."""
from sdm_hip.cli_train import run_training


def main():
    run_training(mode="cold")


if __name__ == "__main__":
    main()
