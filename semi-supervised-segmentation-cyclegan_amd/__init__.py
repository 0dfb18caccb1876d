"""MI355X-native CycleGAN training step.  Import with
`importlib.import_module("semi-supervised-segmentation-cyclegan_amd")` (the directory name is not a
Python identifier)."""
