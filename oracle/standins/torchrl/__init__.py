"""Stand-in for torchrl 0.10.1 (absent): only ``torchrl.modules.MLP`` is used by the reference."""
