"""Train a DDPM/DDIM noise-prediction U-Net on MI355X — same command line (``-c/--config-path``, ``--device``)
and the same JSON config keys, checkpoint files and log lines as the reference's train_diffusion.py.

Differences, all additive:
  * the step body (train_diffusion.py:307-366) runs through sdm_hip.trainer.DiffusionTrainer: fused q-sample,
    hand-written forward/backward kernels, fused Adam(lr, betas=(0.5, 0.999)); fp32 (the reference's fp16
    autocast/GradScaler pair is replaced by exact-fp32 matrix-core kernels, so no loss scaling is needed);
  * launched under ``torchrun`` (WORLD_SIZE > 1) it trains data-parallel: each rank takes a disjoint shard of
    every epoch, gradients are all-reduced over RCCL, rank 0 alone writes checkpoints / plots / logs;
  * ``dataset_path`` may be ``synthetic:N:C:H:W``.
Optional extra config keys (ignored by the reference): ``seed``; ``"train_dtype": "bf16"`` selects the bf16 training path
(bf16 activations / contraction weight copies, fp32 master weights, gradients and Adam).
"""
from sdm_hip.cli_train import run_training


def main():
    run_training(sr=False)


if __name__ == "__main__":
    main()
