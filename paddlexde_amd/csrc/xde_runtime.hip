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

#include <map>
#include <unordered_map>

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
// Idle events, per DEVICE: an event is bound to the device that was current when it was created, and recording it on another
// device's stream fails — a process that times launches on more than one device (the binding allows it) must never be handed a
// foreign one.  `g_event_home` remembers where each event was created.
static std::map<int, std::vector<hipEvent_t>> g_event_pools;
static std::unordered_map<hipEvent_t, int> g_event_home;
double g_prof_bytes[XDE_KID_COUNT] = {0};
int64_t g_prof_counts[XDE_KID_COUNT] = {0};
double g_prof_ms[XDE_KID_COUNT] = {0};

static int current_device() {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) {
    (void)hipGetLastError();
    return -1;  // (no device: events cannot be created either)
  }
  return dev;
}

hipEvent_t get_event() {
  const int dev = current_device();
  auto& pool = g_event_pools[dev];
  if (!pool.empty()) {
    hipEvent_t e = pool.back();
    pool.pop_back();
    return e;
  }
  hipEvent_t e = nullptr;
  if (hipEventCreate(&e) != hipSuccess) {
    (void)hipGetLastError();
    return nullptr;
  }
  g_event_home[e] = dev;
  return e;
}

void put_event(hipEvent_t e) {
  if (!e) return;
  auto it = g_event_home.find(e);
  g_event_pools[it == g_event_home.end() ? -1 : it->second].push_back(e);
}


}  // namespace xde

using namespace xde;

namespace {

// Zero-fill (or pattern-fill) kernel that stands in for a hipGraph MEMSET node: same destination, element size, width,
// height and pitch as the hipMemsetParams it replaces.
__global__ void xde_graph_fill_kernel(unsigned char* dst, unsigned int value, unsigned int elem, size_t width, size_t height,
                                      size_t pitch) {
  const size_t total = width * height;
  for (size_t i = size_t(blockIdx.x) * blockDim.x + threadIdx.x; i < total; i += size_t(gridDim.x) * blockDim.x) {
    const size_t r = i / width, c = i - r * width;
    unsigned char* p = dst + r * pitch + c * elem;
    if (elem == 4)
      *reinterpret_cast<unsigned int*>(p) = value;
    else if (elem == 2)
      *reinterpret_cast<unsigned short*>(p) = static_cast<unsigned short>(value);
    else
      *p = static_cast<unsigned char>(value);
  }
}

}  // namespace

extern "C" {

int xde_graph_replace_memsets(void* graph_v, int* n_replaced) {
  if (!graph_v || !n_replaced) return fail(XDE_EBADARG, "xde_graph_replace_memsets: null pointer");
  hipGraph_t graph = static_cast<hipGraph_t>(graph_v);
  *n_replaced = 0;
  size_t n = 0;
  HIP_TRY(hipGraphGetNodes(graph, nullptr, &n));
  std::vector<hipGraphNode_t> nodes(n);
  if (n) HIP_TRY(hipGraphGetNodes(graph, nodes.data(), &n));
  for (size_t i = 0; i < n; ++i) {
    hipGraphNodeType type;
    HIP_TRY(hipGraphNodeGetType(nodes[i], &type));
    if (type != hipGraphNodeTypeMemset) continue;
    hipMemsetParams mp;
    HIP_TRY(hipGraphMemsetNodeGetParams(nodes[i], &mp));
    if (mp.elementSize != 1 && mp.elementSize != 2 && mp.elementSize != 4)
      return fail(XDE_EBADARG, "xde_graph_replace_memsets: unsupported memset element size");
    size_t nd = 0, nc = 0;
    HIP_TRY(hipGraphNodeGetDependencies(nodes[i], nullptr, &nd));
    std::vector<hipGraphNode_t> deps(nd);
    if (nd) HIP_TRY(hipGraphNodeGetDependencies(nodes[i], deps.data(), &nd));
    HIP_TRY(hipGraphNodeGetDependentNodes(nodes[i], nullptr, &nc));
    std::vector<hipGraphNode_t> children(nc);
    if (nc) HIP_TRY(hipGraphNodeGetDependentNodes(nodes[i], children.data(), &nc));

    unsigned char* dst = static_cast<unsigned char*>(mp.dst);
    unsigned int value = mp.value;
    unsigned int elem = mp.elementSize;
    size_t width = mp.width, height = mp.height ? mp.height : 1, pitch = mp.pitch;
    void* args[] = {&dst, &value, &elem, &width, &height, &pitch};
    const size_t total = width * height;
    unsigned int blocks = static_cast<unsigned int>((total + 255) / 256);
    if (blocks < 1) blocks = 1;
    if (blocks > 1024) blocks = 1024;
    hipKernelNodeParams kp;
    memset(&kp, 0, sizeof(kp));
    kp.func = reinterpret_cast<void*>(xde_graph_fill_kernel);
    kp.gridDim = dim3(blocks);
    kp.blockDim = dim3(256);
    kp.sharedMemBytes = 0;
    kp.kernelParams = args;
    kp.extra = nullptr;
    hipGraphNode_t fill;
    HIP_TRY(hipGraphAddKernelNode(&fill, graph, nd ? deps.data() : nullptr, nd, &kp));
    for (size_t c = 0; c < nc; ++c) HIP_TRY(hipGraphAddDependencies(graph, &fill, &children[c], 1));
    HIP_TRY(hipGraphDestroyNode(nodes[i]));
    ++*n_replaced;
  }
  return XDE_OK;
}

const char* xde_last_error(void) { return g_last_error.c_str(); }
int xde_abi_version(void) { return XDE_ABI_VERSION; }
int64_t xde_sizeof_ctrl(void) { return int64_t(sizeof(xde_ctrl_t)); }
int64_t xde_sizeof_ctrl_params(void) { return int64_t(sizeof(xde_ctrl_params_t)); }
int64_t xde_sizeof_segments(void) { return int64_t(sizeof(xde_segments_t)); }
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
      put_event(r.start);
      put_event(r.stop);
    }
    g_prof_recs.clear();
    // The events sampled launches will be stamped with exist BEFORE the first of them is taken: hipEventCreate costs 1-2 us, and created
    // on demand the first timed block of a benchmark paid for ~100 of them (bench.py's first 20-step block read 7 us per step longer
    // than the two blocks after it).  On a box without a device nothing can be created: the pool stays empty and launches fail anyway.
    // (the pool warmed is the CURRENT device's: the one whose launches the caller is about to time; another device's pool fills on
    // demand)
    constexpr size_t kWarmEvents = 512;
    const int dev = current_device();
    auto& pool = g_event_pools[dev];
    while (dev >= 0 && pool.size() < kWarmEvents) {
      hipEvent_t e;
      if (hipEventCreate(&e) != hipSuccess) {
        (void)hipGetLastError();
        break;
      }
      g_event_home[e] = dev;
      pool.push_back(e);
    }
  }
  return XDE_OK;
}

int xde_prof_collect(int64_t* counts_out, double* ms_out, double* bytes_out) {
  if (!counts_out || !ms_out || !bytes_out) return fail(XDE_EBADARG, "xde_prof_collect: null pointer");
  std::lock_guard<std::mutex> lk(g_prof_mu);
  for (auto& r : g_prof_recs) {
    // a sampled launch that was RECORDED into a hipGraph instead of executed (stream capture) never stamped its events:
    // it is dropped from the statistics, not an error
    float ms = 0.f;
    if (hipEventSynchronize(r.stop) == hipSuccess && hipEventElapsedTime(&ms, r.start, r.stop) == hipSuccess) {
      g_prof_counts[r.kid] += 1;
      g_prof_ms[r.kid] += double(ms);
      g_prof_bytes[r.kid] += r.bytes;
    } else {
      (void)hipGetLastError();
    }
    put_event(r.start);
    put_event(r.stop);
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

