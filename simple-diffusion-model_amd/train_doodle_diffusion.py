"""Train an image-conditioned ("doodle") noise-prediction U-Net on MI355X — same command line and JSON config keys as the
reference's train_doodle_diffusion.py (no ``use_conditional`` / ``flip_imgs`` keys; ``in_channel`` = image + label
channels).

The dataset yields (image, label image) pairs (custom_dataset/doodle_dataset.py); the label image is concatenated to
x_t on the channel axis (train_doodle_diffusion.py:304) and handed to the samplers as ``cond_img`` at checkpoint plots.
The step body is sdm_hip.trainer.DiffusionTrainer.step(..., cond_img=...).  ``dataset_path`` may be
``synthetic-pairs:N:C:H:W``."""
from sdm_hip.cli_train import run_training


def main():
    run_training(mode="doodle")


if __name__ == "__main__":
    main()
