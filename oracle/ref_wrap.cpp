// ref_wrap.cpp — builds the REFERENCE's own CPU kernels, unmodified and from where they lie
// (/root/reference/maskrcnn_benchmark/csrc), into oracle/_ref/detops_ref_C*.so.
//
// TEST INFRASTRUCTURE ONLY (see oracle/README.md).  No reference source is copied into this
// repository: the two translation units are #included by path (-I <reference>/csrc).
//
// The reference was written against PyTorch 1.0; with torch 2.x exactly one construct no longer
// compiles: `AT_DISPATCH_FLOATING_TYPES(tensor.type(), ...)` (cpu/ROIAlign_cpu.cpp:242,
// cpu/nms_cpu.cpp:71) passes a DeprecatedTypeProperties where a ScalarType is now required.
// Instead of editing the sources we re-define the macro so that it accepts both.
#include <torch/extension.h>

namespace detops_ref_shim {
inline at::ScalarType st(const at::DeprecatedTypeProperties& t) { return t.scalarType(); }
inline at::ScalarType st(at::ScalarType t) { return t; }
}  // namespace detops_ref_shim

#undef AT_DISPATCH_FLOATING_TYPES
#define AT_DISPATCH_FLOATING_TYPES(TYPE, NAME, ...) \
  AT_DISPATCH_SWITCH(detops_ref_shim::st(TYPE), NAME, AT_DISPATCH_CASE_FLOATING_TYPES(__VA_ARGS__))

// reference translation units (read-only, compiled in place)
#include "cpu/ROIAlign_cpu.cpp"
#include "cpu/nms_cpu.cpp"
// reference dispatch headers: ROIAlign_forward / nms (CPU branch; WITH_CUDA undefined)
#include "ROIAlign.h"
#include "nms.h"

PYBIND11_MODULE(TORCH_EXTENSION_NAME, m) {
  m.def("nms", &nms, "reference non-maximum suppression (csrc/nms.h:10)");
  m.def("roi_align_forward", &ROIAlign_forward, "reference ROIAlign_forward (csrc/ROIAlign.h:11)");
  m.def("roi_align_backward", &ROIAlign_backward, "reference ROIAlign_backward (raises on CPU)");
}
