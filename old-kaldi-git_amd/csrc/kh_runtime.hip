// kh_runtime.hip — device selection, stream, caching allocator, copies.
// Replaces CuDevice (cudamatrix/cu-device.{h,cc}) for the hot path.
#include <map>
#include <mutex>
#include <unordered_map>
#include <vector>

#include "kh_common.h"

namespace kh {

namespace {
thread_local char g_err[1024] = "";
std::mutex g_mu;
bool g_selected = false;
int g_device = -1;
int g_num_cus = 0;
hipStream_t g_own_stream = nullptr;
hipStream_t g_ext_stream = nullptr;
bool g_use_ext = false;

// Caching pool: the reference allocates on every CuMatrix::Resize
// (cu-matrix.cc:47-100, uncached cu-device.cc:530-555); here freed blocks are
// kept in size-bucketed free lists (power-of-two >= 256 B up to 1 MiB granules,
// then 2 MiB multiples) and reused.
struct Pool {
  std::multimap<size_t, void *> free_blocks;
  std::unordered_map<void *, size_t> live;
  size_t cached_bytes = 0;
} g_pool;

size_t RoundSize(size_t n) {
  if (n < 256) n = 256;
  if (n <= (1u << 20)) {
    size_t p = 256;
    while (p < n) p <<= 1;
    return p;
  }
  const size_t g = 2u << 20;
  return (n + g - 1) / g * g;
}
}  // namespace

void SetError(const char *fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}
const char *LastError() { return g_err; }

static int SelectLocked(int ordinal) {
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess || n <= 0) {
    SetError("no HIP device available (%s); libkaldi_hip has no CPU fallback",
             e == hipSuccess ? "device count is 0" : hipGetErrorString(e));
    return KH_EDEVICE;
  }
  if (ordinal < 0) {
    if (g_selected) return KH_OK;
    ordinal = 0;
  }
  if (ordinal >= n) {
    SetError("kh_select_gpu: ordinal %d out of range (%d devices)", ordinal, n);
    return KH_EINVAL;
  }
  KH_HIP(hipSetDevice(ordinal));
  hipDeviceProp_t prop;
  KH_HIP(hipGetDeviceProperties(&prop, ordinal));
  if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
    SetError("device %d is %s; this library is built for gfx950 only", ordinal,
             prop.gcnArchName);
    return KH_EDEVICE;
  }
  g_num_cus = prop.multiProcessorCount;
  if (g_selected && g_device != ordinal && g_own_stream) {
    g_own_stream = nullptr;  // belongs to the other device; leak on purpose
  }
  if (!g_own_stream)
    // A BLOCKING stream: it orders itself against the legacy default stream, where a host
    // framework that owns the device buffers (torch) enqueues its asynchronous fills and
    // copies - the synchronous semantics the reference's callers expect.
    KH_HIP(hipStreamCreateWithFlags(&g_own_stream, hipStreamDefault));
  g_device = ordinal;
  g_selected = true;
  return KH_OK;
}

int EnsureDevice() {
  if (g_selected) return KH_OK;
  std::lock_guard<std::mutex> l(g_mu);
  return SelectLocked(-1);
}

hipStream_t Stream() { return g_use_ext ? g_ext_stream : g_own_stream; }
int NumCUs() { return g_num_cus > 0 ? g_num_cus : 256; }

void *PoolMalloc(size_t bytes) {
  if (EnsureDevice() != KH_OK) return nullptr;
  size_t sz = RoundSize(bytes);
  std::lock_guard<std::mutex> l(g_mu);
  // best fit: the smallest cached block of at least sz, wasting at most 25 %
  auto it = g_pool.free_blocks.lower_bound(sz);
  void *p = nullptr;
  if (it != g_pool.free_blocks.end() && it->first <= sz + sz / 4) {
    p = it->second;
    sz = it->first;
    g_pool.free_blocks.erase(it);
    g_pool.cached_bytes -= sz;
  } else {
    hipError_t e = hipMalloc(&p, sz);
    if (e != hipSuccess) {
      // give cached blocks back and retry once
      for (auto &kv : g_pool.free_blocks) (void)hipFree(kv.second);
      g_pool.free_blocks.clear();
      g_pool.cached_bytes = 0;
      e = hipMalloc(&p, sz);
      if (e != hipSuccess) {
        SetError("hipMalloc(%zu) failed: %s", sz, hipGetErrorString(e));
        (void)hipGetLastError();
        return nullptr;
      }
      (void)hipGetLastError();  // the first failure is sticky: clear it
    }
  }
  g_pool.live[p] = sz;
  return p;
}

int PoolFree(void *p) {
  if (!p) return KH_OK;
  std::lock_guard<std::mutex> l(g_mu);
  auto it = g_pool.live.find(p);
  if (it == g_pool.live.end()) {
    SetError("kh_free: pointer %p was not allocated by kh_malloc", p);
    return KH_EINVAL;
  }
  size_t sz = it->second;
  g_pool.live.erase(it);
  g_pool.free_blocks.emplace(sz, p);
  g_pool.cached_bytes += sz;
  return KH_OK;
}

size_t PoolCachedBytes() {
  std::lock_guard<std::mutex> l(g_mu);
  return g_pool.cached_bytes;
}

}  // namespace kh

using namespace kh;

extern "C" {

const char *kh_last_error(void) { return LastError(); }

int kh_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

int kh_select_gpu(int ordinal) {
  std::lock_guard<std::mutex> l(g_mu);
  return SelectLocked(ordinal);
}

int kh_enabled(void) { return g_selected ? 1 : 0; }

int kh_device_name(char *buf, size_t len) {
  int rc = EnsureDevice();
  if (rc) return rc;
  hipDeviceProp_t prop;
  KH_HIP(hipGetDeviceProperties(&prop, g_device));
  snprintf(buf, len, "%s (%s)", prop.name, prop.gcnArchName);
  return KH_OK;
}

int kh_mem_info(size_t *free_bytes, size_t *total_bytes) {
  int rc = EnsureDevice();
  if (rc) return rc;
  KH_HIP(hipMemGetInfo(free_bytes, total_bytes));
  return KH_OK;
}

int kh_set_stream(void *hip_stream) {
  int rc = EnsureDevice();
  if (rc) return rc;
  g_ext_stream = static_cast<hipStream_t>(hip_stream);
  g_use_ext = (hip_stream != nullptr);
  return KH_OK;
}

void *kh_get_stream(void) { return static_cast<void *>(Stream()); }

int kh_synchronize(void) {
  int rc = EnsureDevice();
  if (rc) return rc;
  KH_HIP(hipStreamSynchronize(Stream()));
  return KH_OK;
}

void *kh_malloc(size_t bytes) { return PoolMalloc(bytes); }

void *kh_malloc_pitch(size_t row_bytes, size_t num_rows, size_t *pitch_bytes) {
  size_t pitch = (row_bytes + 255) / 256 * 256;
  if (pitch_bytes) *pitch_bytes = pitch;
  return PoolMalloc(pitch * (num_rows ? num_rows : 1));
}

int kh_free(void *ptr) { return PoolFree(ptr); }

int kh_pool_release(void) {
  std::lock_guard<std::mutex> l(g_mu);
  for (auto &kv : g_pool.free_blocks) (void)hipFree(kv.second);
  g_pool.free_blocks.clear();
  g_pool.cached_bytes = 0;
  return KH_OK;
}

int kh_memcpy_2d(void *dst, size_t dst_pitch, const void *src, size_t src_pitch,
                 size_t width_bytes, size_t height, int kind) {
  int rc = EnsureDevice();
  if (rc) return rc;
  KH_CHECK_ARG(kind >= 0 && kind <= 2);
  if (width_bytes == 0 || height == 0) return KH_OK;
  hipMemcpyKind k = kind == 0   ? hipMemcpyHostToDevice
                    : kind == 1 ? hipMemcpyDeviceToHost
                                : hipMemcpyDeviceToDevice;
  KH_HIP(hipMemcpy2DAsync(dst, dst_pitch, src, src_pitch, width_bytes, height, k,
                          Stream()));
  KH_HIP(hipStreamSynchronize(Stream()));
  return KH_OK;
}

int kh_memset(void *dst, int value, size_t bytes) {
  int rc = EnsureDevice();
  if (rc) return rc;
  KH_HIP(hipMemsetAsync(dst, value, bytes, Stream()));
  return KH_OK;
}

}  // extern "C"
