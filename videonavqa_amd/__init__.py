"""videonavqa_amd — MI355X-native implementation of VideoNavQA's video-question fusion path.

Python host code over a C-ABI HIP kernel library (include/vnqa_hip.h, videonavqa_amd/csrc/).
"""
__version__ = "0.1.0"
