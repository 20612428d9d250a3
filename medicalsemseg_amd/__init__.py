"""medicalsemseg_amd -- MI355X-native hot path for 3-D medical semantic segmentation.

Host side mirrors the reference's Python surface (``models.model_builder.build_model``, ``engine.*``,
``utils.arguments.get_args``); the arithmetic runs in hand-written HIP kernels behind the C ABI of
``include/msseg.h`` (``libmsseg_hip.so``).  No CPU fallback exists in this package.
"""
__version__ = "0.1.0"
