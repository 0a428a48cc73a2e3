// Runtime of libtnhip.so: device selection, the in-order stream, a pooled
// device allocator, host<->device copies, HIP events and hipGraph capture.
// One process drives one MI355X (288 GB HBM3E); see include/tnh.h.
#include <stdarg.h>
#include <map>
#include <mutex>
#include <unordered_map>
#include <vector>
#include "tnh_internal.h"

namespace tnh {

static thread_local char g_err[1024] = "";
static int g_device = -1;
static hipStream_t g_stream = nullptr;
static int g_cus = 256;
static bool g_capturing = false;

void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}
hipStream_t stream() { return g_stream; }
bool initialised() { return g_device >= 0; }
bool capturing() { return g_capturing; }
int num_cus() { return g_cus; }
// Workgroups per CU of the grid-stride streaming kernels (permutes, elementwise, reductions) on buffers beyond the
// 256 MB Infinity Cache.  Measured (tools/bw_probe, 1 GiB): a plain read reaches 6.37 TB/s with 4 workgroups of
// 256 threads per CU and 5.4-5.6 with 8-32; a plain copy 5.55 TB/s against 4.6-4.8 -- the narrower the front that
// sweeps through memory, the fewer DRAM pages are open at a time.  TNH_STREAM_WGS overrides (A/B).
int stream_wgs_per_cu(int64_t bytes) {
  static const int env = []() { const char* e = getenv("TNH_STREAM_WGS"); return e ? atoi(e) : 0; }();
  if (env > 0) return env;
  return bytes >= (int64_t(256) << 20) ? 4 : 16;
}

// ---------------------------------------------------------------- block pool
// Blocks are rounded to 512 B (small) or 2 MiB (>= 1 MiB) and recycled through
// exact-size free lists.  All work runs on one in-order stream, so a block
// handed back by tnh_free can be reused immediately: any kernel still reading
// it was queued before the kernel that will next write it.
//
// Graph arenas: every block handed out while a hipGraph is being captured belongs
// to that graph's arena.  Inside the capture a freed arena block may be reused
// (the graph replays the captured order), but it never returns to the general
// pool before tnh_graph_destroy: a replay writes to the captured addresses, so
// nobody else may own them in between.  Blocks still referenced by the caller
// (the outputs of the captured sequence) stay valid across replays.
struct Arena {
  std::multimap<size_t, void*> free_blocks;   // reusable inside the capture
  std::unordered_map<void*, size_t> blocks;   // every block of the arena -> size
};

struct Pool {
  std::mutex mu;
  std::multimap<size_t, void*> free_blocks;
  std::unordered_map<void*, size_t> live;  // ptr -> rounded size
  std::unordered_map<void*, Arena*> arena_of;  // blocks pinned by a graph arena
  Arena* capturing = nullptr;
  std::unordered_map<void*, Arena*> graphs;    // hipGraphExec_t -> arena
  int64_t in_use = 0, cached = 0, peak = 0;

  static size_t round(size_t n) {
    if (n == 0) n = 1;
    if (n < (1u << 20)) return (n + 511) & ~size_t(511);
    const size_t g = size_t(2) << 20;
    return (n + g - 1) / g * g;
  }
  void release_cached() {
    for (auto& kv : free_blocks) (void)hipFree(kv.second);
    free_blocks.clear();
    cached = 0;
  }
};
static Pool g_pool;

}  // namespace tnh

using namespace tnh;

extern "C" {

const char* tnh_last_error(void) { return g_err; }
const char* tnh_version(void) { return "tnhip 0.1 (gfx950)"; }

int tnh_device_count(int* count) {
  TNH_REQUIRE(count != nullptr, "count is null");
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess) {
    (void)hipGetLastError();
    n = 0;
  }
  *count = n;
  return TNH_OK;
}

int tnh_init(int device) {
  if (g_device == device && g_stream != nullptr) return TNH_OK;
  TNH_REQUIRE(g_device < 0, "tnh_init(%d): already initialised on device %d",
              device, g_device);
  int n = 0;
  TNH_HIP(hipGetDeviceCount(&n));
  TNH_REQUIRE(device >= 0 && device < n, "device %d out of range (%d visible)",
              device, n);
  TNH_HIP(hipSetDevice(device));
  hipDeviceProp_t prop;
  TNH_HIP(hipGetDeviceProperties(&prop, device));
  if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
    set_error("libtnhip is built for gfx950 only; device %d is %s", device,
              prop.gcnArchName);
    return TNH_ERR_UNSUPPORTED;
  }
  g_cus = prop.multiProcessorCount;
  TNH_HIP(hipStreamCreateWithFlags(&g_stream, hipStreamDefault));
  g_device = device;
  return TNH_OK;
}

int tnh_shutdown(void) {
  if (g_device < 0) return TNH_OK;
  (void)hipStreamSynchronize(g_stream);
  {
    std::lock_guard<std::mutex> lk(g_pool.mu);
    g_pool.release_cached();
  }
  (void)hipStreamDestroy(g_stream);
  g_stream = nullptr;
  g_device = -1;
  return TNH_OK;
}

int tnh_device_info(char* name, int len, int* cus, int64_t* hbm_bytes) {
  TNH_NEED_INIT();
  hipDeviceProp_t prop;
  TNH_HIP(hipGetDeviceProperties(&prop, g_device));
  if (name && len > 0) snprintf(name, len, "%s (%s)", prop.name, prop.gcnArchName);
  if (cus) *cus = prop.multiProcessorCount;
  if (hbm_bytes) *hbm_bytes = (int64_t)prop.totalGlobalMem;
  return TNH_OK;
}

int tnh_device_pci_bus_id(char* buf, int len) {
  TNH_NEED_INIT();
  TNH_REQUIRE(buf != nullptr && len >= 16, "tnh_device_pci_bus_id: buffer of >= 16 bytes needed");
  TNH_HIP(hipDeviceGetPCIBusId(buf, len, g_device));
  return TNH_OK;
}

int tnh_malloc(void** ptr, size_t nbytes) {
  TNH_NEED_INIT();
  TNH_REQUIRE(ptr != nullptr, "ptr is null");
  const size_t sz = Pool::round(nbytes);
  std::lock_guard<std::mutex> lk(g_pool.mu);
  void* p = nullptr;
  Arena* ar = g_pool.capturing;
  if (ar != nullptr) {
    auto ia = ar->free_blocks.find(sz);
    if (ia != ar->free_blocks.end()) {
      p = ia->second;
      ar->free_blocks.erase(ia);
      g_pool.live[p] = sz;
      g_pool.in_use += (int64_t)sz;
      if (g_pool.in_use > g_pool.peak) g_pool.peak = g_pool.in_use;
      *ptr = p;
      return TNH_OK;
    }
  }
  auto it = g_pool.free_blocks.find(sz);
  if (it != g_pool.free_blocks.end()) {
    p = it->second;
    g_pool.free_blocks.erase(it);
    g_pool.cached -= (int64_t)sz;
  } else {
    // (relaxed capture mode: hipMalloc is legal while the stream is capturing)
    hipError_t e = hipMalloc(&p, sz);
    if (e != hipSuccess && ar == nullptr) {
      // retry after returning the cached blocks to the driver.  Never while a capture is open:
      // synchronising a capturing stream is illegal and would invalidate the capture -- the caller
      // gets TNH_ERR_NOMEM and falls back to the eager path (distributed.contract_sliced).
      (void)hipGetLastError();
      (void)hipStreamSynchronize(g_stream);
      g_pool.release_cached();
      e = hipMalloc(&p, sz);
    }
    if (e != hipSuccess) {
      (void)hipGetLastError();
      set_error("out of device memory allocating %zu bytes (in use %lld)", sz,
                (long long)g_pool.in_use);
      return TNH_ERR_NOMEM;
    }
  }
  if (ar != nullptr) {
    ar->blocks[p] = sz;
    g_pool.arena_of[p] = ar;
  }
  g_pool.live[p] = sz;
  g_pool.in_use += (int64_t)sz;
  if (g_pool.in_use > g_pool.peak) g_pool.peak = g_pool.in_use;
  *ptr = p;
  return TNH_OK;
}

int tnh_free(void* ptr) {
  if (ptr == nullptr) return TNH_OK;
  if (g_device < 0) return TNH_OK;  // after shutdown: nothing to do
  std::lock_guard<std::mutex> lk(g_pool.mu);
  auto it = g_pool.live.find(ptr);
  TNH_REQUIRE(it != g_pool.live.end(), "tnh_free(%p): not a live block", ptr);
  const size_t sz = it->second;
  g_pool.live.erase(it);
  g_pool.in_use -= (int64_t)sz;
  auto ia = g_pool.arena_of.find(ptr);
  if (ia != g_pool.arena_of.end()) {
    // pinned by a graph: reusable only inside that graph's own capture
    if (ia->second == g_pool.capturing) ia->second->free_blocks.emplace(sz, ptr);
    return TNH_OK;
  }
  g_pool.free_blocks.emplace(sz, ptr);
  g_pool.cached += (int64_t)sz;
  return TNH_OK;
}

int tnh_trim(void) {
  TNH_NEED_INIT();
  TNH_HIP(hipStreamSynchronize(g_stream));
  std::lock_guard<std::mutex> lk(g_pool.mu);
  g_pool.release_cached();
  return TNH_OK;
}

int tnh_mem_stats(int64_t* in_use, int64_t* cached, int64_t* peak) {
  std::lock_guard<std::mutex> lk(g_pool.mu);
  if (in_use) *in_use = g_pool.in_use;
  if (cached) *cached = g_pool.cached;
  if (peak) *peak = g_pool.peak;
  return TNH_OK;
}

int tnh_pool_has(size_t nbytes, int* has) {
  TNH_REQUIRE(has != nullptr, "has is null");
  const size_t sz = Pool::round(nbytes);
  std::lock_guard<std::mutex> lk(g_pool.mu);
  Arena* ar = g_pool.capturing;
  *has = ((ar != nullptr && ar->free_blocks.find(sz) != ar->free_blocks.end()) ||
          g_pool.free_blocks.find(sz) != g_pool.free_blocks.end()) ? 1 : 0;
  return TNH_OK;
}

int tnh_h2d(void* dst, const void* host_src, size_t nbytes) {
  TNH_NEED_INIT();
  if (nbytes == 0) return TNH_OK;
  TNH_REQUIRE(dst && host_src, "null pointer");
  // Pageable source: hipMemcpyAsync stages it before returning, so the host
  // buffer may be reused by the caller right away.
  TNH_HIP(hipMemcpyAsync(dst, host_src, nbytes, hipMemcpyHostToDevice, g_stream));
  TNH_HIP(hipStreamSynchronize(g_stream));
  return TNH_OK;
}

int tnh_d2h(void* host_dst, const void* src, size_t nbytes) {
  TNH_NEED_INIT();
  if (nbytes == 0) return TNH_OK;
  TNH_REQUIRE(host_dst && src, "null pointer");
  TNH_HIP(hipMemcpyAsync(host_dst, src, nbytes, hipMemcpyDeviceToHost, g_stream));
  TNH_HIP(hipStreamSynchronize(g_stream));
  return TNH_OK;
}

int tnh_d2d(void* dst, const void* src, size_t nbytes) {
  TNH_NEED_INIT();
  if (nbytes == 0) return TNH_OK;
  TNH_REQUIRE(dst && src, "null pointer");
  TNH_HIP(hipMemcpyAsync(dst, src, nbytes, hipMemcpyDeviceToDevice, g_stream));
  return TNH_OK;
}

int tnh_memset(void* dst, int byte, size_t nbytes) {
  TNH_NEED_INIT();
  if (nbytes == 0) return TNH_OK;
  TNH_REQUIRE(dst != nullptr, "null pointer");
  TNH_HIP(hipMemsetAsync(dst, byte, nbytes, g_stream));
  return TNH_OK;
}

int tnh_sync(void) {
  TNH_NEED_INIT();
  TNH_HIP(hipStreamSynchronize(g_stream));
  return TNH_OK;
}

int tnh_stream(void** s) {
  TNH_NEED_INIT();
  TNH_REQUIRE(s != nullptr, "null pointer");
  *s = (void*)g_stream;
  return TNH_OK;
}

// ----------------------------------------------------------------- events
int tnh_event_create(void** ev) {
  TNH_NEED_INIT();
  TNH_REQUIRE(ev != nullptr, "null pointer");
  hipEvent_t e;
  TNH_HIP(hipEventCreate(&e));
  *ev = (void*)e;
  return TNH_OK;
}
int tnh_event_record(void* ev) {
  TNH_NEED_INIT();
  TNH_HIP(hipEventRecord((hipEvent_t)ev, g_stream));
  return TNH_OK;
}
int tnh_event_sync(void* ev) {
  TNH_NEED_INIT();
  TNH_HIP(hipEventSynchronize((hipEvent_t)ev));
  return TNH_OK;
}
int tnh_event_elapsed_ms(void* start, void* stop, float* ms) {
  TNH_NEED_INIT();
  TNH_REQUIRE(ms != nullptr, "null pointer");
  TNH_HIP(hipEventElapsedTime(ms, (hipEvent_t)start, (hipEvent_t)stop));
  return TNH_OK;
}
int tnh_event_destroy(void* ev) {
  if (ev == nullptr) return TNH_OK;
  TNH_HIP(hipEventDestroy((hipEvent_t)ev));
  return TNH_OK;
}

// ----------------------------------------------------------------- graphs
int tnh_graph_begin(void) {
  TNH_NEED_INIT();
  TNH_REQUIRE(!g_capturing, "graph capture already in progress");
  TNH_HIP(hipStreamBeginCapture(g_stream, hipStreamCaptureModeRelaxed));
  g_capturing = true;
  std::lock_guard<std::mutex> lk(g_pool.mu);
  g_pool.capturing = new Arena();
  return TNH_OK;
}
int tnh_graph_end(void** graph_exec) {
  TNH_NEED_INIT();
  TNH_REQUIRE(g_capturing, "no graph capture in progress");
  TNH_REQUIRE(graph_exec != nullptr, "null pointer");
  g_capturing = false;
  Arena* ar = nullptr;
  {
    std::lock_guard<std::mutex> lk(g_pool.mu);
    ar = g_pool.capturing;
    g_pool.capturing = nullptr;
  }
  auto drop_arena = [&]() {
    std::lock_guard<std::mutex> lk(g_pool.mu);
    for (auto& kv : ar->blocks) {
      g_pool.arena_of.erase(kv.first);
      if (g_pool.live.find(kv.first) == g_pool.live.end()) {
        g_pool.free_blocks.emplace(kv.second, kv.first);
        g_pool.cached += (int64_t)kv.second;
      }
    }
    delete ar;
  };
  hipGraph_t graph = nullptr;
  hipError_t e = hipStreamEndCapture(g_stream, &graph);
  if (e != hipSuccess) {
    (void)hipGetLastError();
    drop_arena();
    set_error("hipStreamEndCapture failed: %s", hipGetErrorString(e));
    return TNH_ERR_HIP;
  }
  hipGraphExec_t exec = nullptr;
  e = hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0);
  (void)hipGraphDestroy(graph);
  if (e != hipSuccess) {
    (void)hipGetLastError();
    drop_arena();
    set_error("hipGraphInstantiate failed: %s", hipGetErrorString(e));
    return TNH_ERR_HIP;
  }
  {
    std::lock_guard<std::mutex> lk(g_pool.mu);
    g_pool.graphs[(void*)exec] = ar;
  }
  *graph_exec = (void*)exec;
  return TNH_OK;
}
int tnh_graph_launch(void* graph_exec) {
  TNH_NEED_INIT();
  TNH_HIP(hipGraphLaunch((hipGraphExec_t)graph_exec, g_stream));
  return TNH_OK;
}
int tnh_graph_destroy(void* graph_exec) {
  if (graph_exec == nullptr) return TNH_OK;
  if (g_device < 0) return TNH_OK;
  // replays still in flight read/write the arena: drain before the blocks go back
  TNH_HIP(hipStreamSynchronize(g_stream));
  TNH_HIP(hipGraphExecDestroy((hipGraphExec_t)graph_exec));
  std::lock_guard<std::mutex> lk(g_pool.mu);
  auto it = g_pool.graphs.find(graph_exec);
  if (it != g_pool.graphs.end()) {
    Arena* ar = it->second;
    for (auto& kv : ar->blocks) {
      g_pool.arena_of.erase(kv.first);
      if (g_pool.live.find(kv.first) == g_pool.live.end()) {  // not referenced by the caller any more
        g_pool.free_blocks.emplace(kv.second, kv.first);
        g_pool.cached += (int64_t)kv.second;
      }
    }
    delete ar;
    g_pool.graphs.erase(it);
  }
  return TNH_OK;
}

}  // extern "C"
