// version.hip — library identification.
#include "detops_common.h"

DETOPS_API int detops_version(const char** arch) {
  if (arch) *arch = "gfx950";
  return DETOPS_ABI_VERSION;
}
