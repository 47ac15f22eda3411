"""3dbodyanimation_amd — MI355X-native SMPL residual/Jacobian evaluator.

The directory name starts with a digit, so import it with
``importlib.import_module("3dbodyanimation_amd")``.
"""
