// tuning.hip — the library's few tuning / test switches (host code only).  Set once at load from the environment
// variable DETOPS_TUNING="key=value,key=value" or at run time through detops_tuning_set (tests, A/B measurements);
// the launch paths only read the struct.
#include <cstdlib>
#include <cstring>

#include "detops_common.h"

namespace {
struct Key { const char* name; int DetopsTuning::*field; };
const Key kKeys[] = {
    {"roi_bwd_impl", &DetopsTuning::roi_bwd_impl},       {"roi_bwd_seg", &DetopsTuning::roi_bwd_seg},
    {"roi_bwd_ring", &DetopsTuning::roi_bwd_ring},       {"roi_bwd_ct", &DetopsTuning::roi_bwd_ct},
    {"roi_bwd_groups", &DetopsTuning::roi_bwd_groups},   {"roi_bwd_scan_ct", &DetopsTuning::roi_bwd_scan_ct},
    {"roi_bwd_debug", &DetopsTuning::roi_bwd_debug},     {"roi_fwd_impl", &DetopsTuning::roi_fwd_impl},
    {"roi_fwd_order", &DetopsTuning::roi_fwd_order},     {"roi_fwd_order_mink", &DetopsTuning::roi_fwd_order_mink},
    {"dcn_col2im", &DetopsTuning::dcn_col2im},           {"dcn_fused", &DetopsTuning::dcn_fused},
    {"dcn_gather_xcd", &DetopsTuning::dcn_gather_xcd},   {"dcn_nhwc", &DetopsTuning::dcn_nhwc},
    {"nms_fused", &DetopsTuning::nms_fused},             {"roi_fwd_records", &DetopsTuning::roi_fwd_records},
    {"roi_fwd_ct", &DetopsTuning::roi_fwd_ct},           {"dcn_ell_build", &DetopsTuning::dcn_ell_build},
    {"nms_fault", &DetopsTuning::nms_fault},             {"nms_spin_budget", &DetopsTuning::nms_spin_budget},
    {"roi_bwd_split", &DetopsTuning::roi_bwd_split},     {"roi_bwd_maxseg", &DetopsTuning::roi_bwd_maxseg},
    {"roi_bwd_extras", &DetopsTuning::roi_bwd_extras},   {"nms_no_repair", &DetopsTuning::nms_no_repair},
    {"nms_no_presorted", &DetopsTuning::nms_no_presorted}, {"nms_debug", &DetopsTuning::nms_debug},
};

bool set_key(DetopsTuning& t, const char* key, size_t len, int value) {
  for (const Key& k : kKeys)
    if (strlen(k.name) == len && strncmp(k.name, key, len) == 0) { t.*(k.field) = value; return true; }
  return false;
}

DetopsTuning from_env() {
  DetopsTuning t{};
  const char* e = getenv("DETOPS_TUNING");
  while (e && *e) {
    const char* eq = strchr(e, '=');
    if (!eq) break;
    set_key(t, e, static_cast<size_t>(eq - e), atoi(eq + 1));
    const char* comma = strchr(eq, ',');
    e = comma ? comma + 1 : nullptr;
  }
  return t;
}
}  // namespace

DetopsTuning& detops_tuning() {
  static DetopsTuning t = from_env();
  return t;
}

DETOPS_API int detops_tuning_set(const char* key, int value) {
  if (!key) return DETOPS_EINVAL;
  return set_key(detops_tuning(), key, strlen(key), value) ? 0 : DETOPS_EINVAL;
}

DETOPS_API int detops_tuning_get(const char* key, int* value) {
  if (!key || !value) return DETOPS_EINVAL;
  for (const Key& k : kKeys)
    if (strcmp(k.name, key) == 0) { *value = detops_tuning().*(k.field); return 0; }
  return DETOPS_EINVAL;
}
