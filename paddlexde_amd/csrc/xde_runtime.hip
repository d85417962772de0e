// libxde_hip.so — hand-written gfx950 (MI355X / CDNA4) kernels for the Runge–Kutta hot path of
// paddlexde_amd.  C ABI: include/xde_hip.h (each entry point cites the reference lines it replaces).
//
// All kernels are HBM-bandwidth bound (≈0.5 flop/byte): 16-byte-per-lane coalesced loads, grid-stride
// loops sized to keep every CU's memory queue full (2048 workgroups × 256 threads = 8 waves/SIMD),
// fp32 per-thread accumulation → wave64 __shfl_down → LDS cross-wave → one fp64 partial per workgroup,
// reduced in a fixed order by a single workgroup (bit-reproducible, and identical on every rank after
// the all-reduce).  No MFMA: there is no contraction on this path.
//
// Built with -ffp-contract=off so that element-wise results follow the reference's (unfused) op
// order exactly; the kernels are memory bound, so FMA contraction would buy nothing.
//
// Library state: last error, launch-sampling records, ABI housekeeping.

#include "xde_common.hpp"

namespace xde {

thread_local std::string g_last_error;

int fail(int code, const std::string& msg) {
  g_last_error = msg;
  return code;
}


std::mutex g_prof_mu;
bool g_prof_on = false;
int g_prof_period = 1;
int64_t g_prof_launches[XDE_KID_COUNT] = {0};
std::vector<ProfRec> g_prof_recs;
std::vector<hipEvent_t> g_event_pool;
double g_prof_bytes[XDE_KID_COUNT] = {0};
int64_t g_prof_counts[XDE_KID_COUNT] = {0};
double g_prof_ms[XDE_KID_COUNT] = {0};

hipEvent_t get_event() {
  if (!g_event_pool.empty()) {
    hipEvent_t e = g_event_pool.back();
    g_event_pool.pop_back();
    return e;
  }
  hipEvent_t e;
  (void)hipEventCreate(&e);
  return e;
}


}  // namespace xde

using namespace xde;

extern "C" {

const char* xde_last_error(void) { return g_last_error.c_str(); }
int xde_abi_version(void) { return XDE_ABI_VERSION; }
int64_t xde_sizeof_ctrl(void) { return int64_t(sizeof(xde_ctrl_t)); }
int64_t xde_workspace_bytes(void) { return int64_t(sizeof(NormSlot)) * kSlots; }

int xde_prof_enable(int on) {
  std::lock_guard<std::mutex> lk(g_prof_mu);
  g_prof_on = on != 0;
  g_prof_period = on > 1 ? on : 1;
  if (on) {
    for (int i = 0; i < XDE_KID_COUNT; ++i) {
      g_prof_launches[i] = 0;
      g_prof_bytes[i] = 0;
      g_prof_counts[i] = 0;
      g_prof_ms[i] = 0;
    }
    for (auto& r : g_prof_recs) {
      g_event_pool.push_back(r.start);
      g_event_pool.push_back(r.stop);
    }
    g_prof_recs.clear();
  }
  return XDE_OK;
}

int xde_prof_collect(int64_t* counts_out, double* ms_out, double* bytes_out) {
  if (!counts_out || !ms_out || !bytes_out) return fail(XDE_EBADARG, "xde_prof_collect: null pointer");
  std::lock_guard<std::mutex> lk(g_prof_mu);
  for (auto& r : g_prof_recs) {
    HIP_TRY(hipEventSynchronize(r.stop));
    float ms = 0.f;
    HIP_TRY(hipEventElapsedTime(&ms, r.start, r.stop));
    g_prof_counts[r.kid] += 1;
    g_prof_ms[r.kid] += double(ms);
    g_event_pool.push_back(r.start);
    g_event_pool.push_back(r.stop);
  }
  g_prof_recs.clear();
  for (int i = 0; i < XDE_KID_COUNT; ++i) {
    counts_out[i] = g_prof_counts[i];
    ms_out[i] = g_prof_ms[i];
    bytes_out[i] = g_prof_bytes[i];
  }
  return XDE_OK;
}

}  // extern "C"

