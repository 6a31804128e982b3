"""MI355X-native IntEL ensemble-scoring engine (hot path of JiayuLi-997/IntEL-SIGIR2023).

Host side = this package (Python, mirrors the reference's model / loss / runner interface);
device side = libintel_hip.so (hand-written gfx950 HIP kernels behind the C ABI of include/intel_hip.h).
"""
__version__ = '0.1.0'
