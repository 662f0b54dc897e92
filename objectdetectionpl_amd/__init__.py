"""objectdetectionpl_amd -- MI355X-native detector math behind the ObjectDetectionPL API.

Only the hot path of the reference lives here: anchor/prior tables, box decode, IoU + target
assignment, the losses of `LightningFunc/losses.py`, batched NMS and the data-parallel gradient
exchange.  All device arithmetic runs in hand-written HIP kernels (csrc/, built for gfx950 into
csrc/libodpl_hip.so, C ABI in include/odpl_hip.h); this package is the Python host that mirrors
the reference's names and signatures.  There is no CPU fallback: a missing library raises.
"""
__version__ = "0.1.0"
