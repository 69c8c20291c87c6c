"""maskrcnn_benchmark — MI355X-native drop-in for the detection-head + data-parallel training path
of facebookresearch/maskrcnn-benchmark (see DESIGN.md / INTEGRATION.md at the repository root).

`maskrcnn_benchmark._C` and `maskrcnn_benchmark.layers` keep the reference's operator API; the
operators themselves are hand-written HIP kernels for gfx950 behind the C ABI in include/detops.h.
"""
