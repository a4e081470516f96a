// Run-time switches of librecon_hip.so: ONE read-only table, filled from the environment the first time anything asks ("optional
// read-only kernel-config table"), instead of getenv() calls scattered through the launch paths (round 4 had ~30 of them, several per call).
// Every switch selects between kernels that compute the same result (A/B measurements, tests that pin a form); none changes semantics.
//   recon_config_set(name, value | NULL)   overrides / clears one entry — between launches, from the thread that launches (tests, tools)
//   recon_config_get(name)                 the current value or NULL
// Names are the former environment variables (RECON_GEMM_CFG ...); INTEGRATION.md lists them.
#include <mutex>
#include <stdlib.h>
#include <string.h>
#include "recon_common.h"

namespace recon {
namespace {
const char* const kNames[CFG_COUNT] = {
    "RECON_BGEMM_CFG", "RECON_GCN_FUSED", "RECON_GCN_FUSED_BWD", "RECON_GCN_FUSED_PARTS", "RECON_GCN_STACK_PARTS", "RECON_GEMM_CFG", "RECON_GEMM_LIN",
    "RECON_GEMM_SPLITK", "RECON_GEMM_XCD", "RECON_PROP_B16", "RECON_PROP_B16_YPOST", "RECON_PROP_BWD", "RECON_PROP_BWD_CHAIN", "RECON_PROP_BWD_WIDE",
    "RECON_PROP_FWD", "RECON_PROP_LDS_KB", "RECON_ATP_ROW_SCALE", "RECON_GRAPH_SMALL", "RECON_GCN_STACK_GPW", "RECON_KG_NHOP", "RECON_HX2_RING", "RECON_K2_LDS_RING", "RECON_K2_PERSIST"};
char g_val[CFG_COUNT][32];
bool g_set[CFG_COUNT];
std::once_flag g_once;

void store(int k, const char* v) {
    g_set[k] = v != nullptr;
    if (v) { strncpy(g_val[k], v, sizeof(g_val[k]) - 1); g_val[k][sizeof(g_val[k]) - 1] = '\0'; }
}
void load_env() {
    for (int k = 0; k < CFG_COUNT; ++k) store(k, getenv(kNames[k]));
}
int key_of(const char* name) {
    if (!name) return -1;
    for (int k = 0; k < CFG_COUNT; ++k) if (strcmp(name, kNames[k]) == 0) return k;
    return -1;
}
}  // namespace

const char* cfg(CfgKey k) {
    std::call_once(g_once, load_env);
    return g_set[k] ? g_val[k] : nullptr;
}
int cfg_int(CfgKey k, int dflt) { const char* v = cfg(k); return v ? atoi(v) : dflt; }
char cfg_char(CfgKey k) { const char* v = cfg(k); return v ? v[0] : '\0'; }
}  // namespace recon

// ---- the optional NaN flag (the reference asserts `not torch.isnan(...).any()` on edge_e, e_rowsum and h_prime, GAT/layers.py:147,167,172:
// three device -> host round trips per layer call).  Here the attention kernels raise ONE device word per device when a row sum comes out
// NaN / infinite (which is what every one of those asserts comes down to: the weights' sum, and Inf / Inf in the division); nobody waits
// for it — the caller reads the word when it likes (recon_amd.gat_layers.nan_raised(), every call under RECON_DEBUG_NAN=1).
namespace recon {
namespace {
int32_t* g_nan[16] = {};
bool g_nan_any = false;
}  // namespace
int32_t* nan_flag() {
    if (!g_nan_any) return nullptr;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return nullptr;
    return g_nan[dev & 15];
}
}  // namespace recon

extern "C" int recon_set_nan_flag(int32_t device, int32_t* flag) {
    if (device < 0 || device >= 16) return RECON_ERR_INVALID;
    recon::g_nan[device] = flag;
    recon::g_nan_any = false;
    for (int i = 0; i < 16; ++i) recon::g_nan_any = recon::g_nan_any || recon::g_nan[i] != nullptr;
    return RECON_OK;
}

extern "C" int recon_config_set(const char* name, const char* value) {
    std::call_once(recon::g_once, recon::load_env);
    const int k = recon::key_of(name);
    if (k < 0) return RECON_ERR_INVALID;
    recon::store(k, value);
    return RECON_OK;
}

extern "C" const char* recon_config_get(const char* name) {
    const int k = recon::key_of(name);
    return k < 0 ? nullptr : recon::cfg(static_cast<recon::CfgKey>(k));
}
