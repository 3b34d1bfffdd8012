"""aesrc2020_amd: MI355X-native SAR-Net train-step hot path."""
