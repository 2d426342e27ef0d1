// Device-wide sort / scan / select primitives (rocPRIM) behind plain functions, error slot,
// and the small id-manipulation kernels of the masking path.
#include <stdarg.h>
#include <string.h>

#include <cstring>

#include <rocprim/rocprim.hpp>

#include "ag_common.hpp"

std::string& ag_error_slot() {
  static thread_local std::string slot;
  return slot;
}

int ag_fail(int code, const char* fmt, ...) {
  char buf[1024];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof(buf), fmt, ap);
  va_end(ap);
  ag_error_slot() = buf;
  return code;
}

// ---------------------------------------------------------------------------------------------
// event profiler
// ---------------------------------------------------------------------------------------------
#include <map>
#include <mutex>
#include <vector>

namespace {
struct ProfRec {
  const char* name;
  hipEvent_t start, stop;
};
std::mutex g_prof_mutex;
bool g_prof_on = false;
bool g_prof_open = false;   // the last begin recorded a start event
std::string g_prof_filter;  // empty: every kernel; otherwise only the kernel of this name
std::vector<ProfRec> g_prof_recs;
std::vector<hipEvent_t> g_prof_pool;

hipEvent_t prof_event() {
  if (!g_prof_pool.empty()) {
    hipEvent_t e = g_prof_pool.back();
    g_prof_pool.pop_back();
    return e;
  }
  hipEvent_t e;
  if (hipEventCreate(&e) != hipSuccess) return nullptr;
  return e;
}
}  // namespace

void ag_prof_begin(const char* name, hipStream_t stream) {
  if (!g_prof_on) return;
  std::lock_guard<std::mutex> lock(g_prof_mutex);
  g_prof_open = g_prof_filter.empty() || g_prof_filter == name;
  if (!g_prof_open) return;
  ProfRec r{name, prof_event(), prof_event()};
  if (r.start) (void)hipEventRecord(r.start, stream);
  g_prof_recs.push_back(r);
}

void ag_prof_end(hipStream_t stream) {
  if (!g_prof_on) return;
  std::lock_guard<std::mutex> lock(g_prof_mutex);
  if (!g_prof_open) return;
  g_prof_open = false;
  if (!g_prof_recs.empty() && g_prof_recs.back().stop) (void)hipEventRecord(g_prof_recs.back().stop, stream);
}

extern "C" int ag_profile_enable(int on) {
  std::lock_guard<std::mutex> lock(g_prof_mutex);
  for (auto& r : g_prof_recs) {
    if (r.start) g_prof_pool.push_back(r.start);
    if (r.stop) g_prof_pool.push_back(r.stop);
  }
  g_prof_recs.clear();
  g_prof_on = on != 0;
  return AG_OK;
}

extern "C" int ag_profile_filter(const char* name) {
  std::lock_guard<std::mutex> lock(g_prof_mutex);
  g_prof_filter = name ? name : "";
  return AG_OK;
}

extern "C" int ag_profile_collect(char* buf, size_t buf_bytes) {
  std::lock_guard<std::mutex> lock(g_prof_mutex);
  std::map<std::string, std::pair<long, double>> agg;
  for (auto& r : g_prof_recs) {
    if (!r.start || !r.stop) continue;
    if (hipEventSynchronize(r.stop) != hipSuccess) continue;
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, r.start, r.stop) != hipSuccess) continue;
    auto& a = agg[r.name];
    a.first += 1;
    a.second += ms;
  }
  std::string out;
  for (auto& kv : agg) {
    char line[256];
    snprintf(line, sizeof(line), "%s %ld %.6f\n", kv.first.c_str(), kv.second.first, kv.second.second);
    out += line;
  }
  if (buf && buf_bytes) {
    snprintf(buf, buf_bytes, "%s", out.c_str());
  }
  return static_cast<int>(out.size());
}

extern "C" const char* ag_last_error(void) { return ag_error_slot().c_str(); }
extern "C" int ag_version(void) { return 100; }

// ---------------------------------------------------------------------------------------------
size_t ag_sort_u64_temp_bytes(int64_t n) {
  size_t bytes = 0;
  uint64_t* p = nullptr;
  (void)rocprim::radix_sort_keys(nullptr, bytes, p, p, static_cast<size_t>(n > 0 ? n : 1), 0u, 64u, hipStream_t(0));
  return bytes + 256;
}

int ag_sort_u64(void* d_temp, size_t temp_bytes, const uint64_t* d_in, uint64_t* d_out, int64_t n, int begin_bit,
                int end_bit, hipStream_t stream) {
  if (n <= 0) return AG_OK;
  // rocPRIM's merge-sort path builds its bit mask as (1 << end_bit) - 1: end_bit == 64 is only
  // safe together with begin_bit == 0 (no mask)
  if (end_bit >= 64 && begin_bit > 0) return ag_fail(AG_ERR_INVALID, "ag_sort_u64: partial sort must end below bit 64");
  if (end_bit <= begin_bit) {
    AG_HIP(hipMemcpyAsync(d_out, d_in, n * sizeof(uint64_t), hipMemcpyDeviceToDevice, stream));
    return AG_OK;
  }
  AgProfScope prof("rocprim_radix_sort_keys", stream);
  AG_HIP(rocprim::radix_sort_keys(d_temp, temp_bytes, d_in, d_out, static_cast<size_t>(n),
                                  static_cast<unsigned>(begin_bit), static_cast<unsigned>(end_bit), stream));
  return AG_OK;
}

size_t ag_sort_pairs_u64_i32_temp_bytes(int64_t n) {
  size_t bytes = 0;
  uint64_t* k = nullptr;
  int32_t* v = nullptr;
  (void)rocprim::radix_sort_pairs(nullptr, bytes, k, k, v, v, static_cast<size_t>(n > 0 ? n : 1), 0u, 64u, hipStream_t(0));
  return bytes + 256;
}

// stable ascending sort of (key, value) pairs on all 64 key bits
int ag_sort_pairs_u64_i32(void* d_temp, size_t temp_bytes, const uint64_t* d_keys_in, uint64_t* d_keys_out,
                          const int32_t* d_vals_in, int32_t* d_vals_out, int64_t n, hipStream_t stream) {
  if (n <= 0) return AG_OK;
  AgProfScope prof("rocprim_radix_sort_pairs", stream);
  AG_HIP(rocprim::radix_sort_pairs(d_temp, temp_bytes, d_keys_in, d_keys_out, d_vals_in, d_vals_out,
                                   static_cast<size_t>(n), 0u, 64u, stream));
  return AG_OK;
}

// Sorts every segment [begin[i], end[i]) of int32 keys ascending on bits [0, end_bit); the keys live
// in one of two buffers of `size` elements and *sorted_in_b tells where the result is.
size_t ag_segmented_sort_i32_temp_bytes(int64_t segments) {
  size_t bytes = 0;
  int32_t* p = nullptr;
  rocprim::double_buffer<int32_t> keys(p, p);
  const uint32_t* o = nullptr;
  (void)rocprim::segmented_radix_sort_keys(nullptr, bytes, keys, 1u << 30, static_cast<unsigned>(segments > 0 ? segments : 1),
                                           o, o, 0u, 32u, hipStream_t(0));
  return bytes + 256;
}

int ag_segmented_sort_i32(void* d_temp, size_t temp_bytes, int32_t* d_a, int32_t* d_b, int64_t size, int64_t segments,
                          const uint32_t* d_begin, const uint32_t* d_end, int end_bit, bool* sorted_in_b,
                          hipStream_t stream) {
  if (size >= (int64_t(1) << 32) || segments >= (int64_t(1) << 32))
    return ag_fail(AG_ERR_INVALID, "segmented sort: more than 2^32 items");
  rocprim::double_buffer<int32_t> keys(d_a, d_b);
  AgProfScope prof("rocprim_segmented_radix_sort", stream);
  AG_HIP(rocprim::segmented_radix_sort_keys(d_temp, temp_bytes, keys, static_cast<unsigned>(size),
                                            static_cast<unsigned>(segments), d_begin, d_end, 0u,
                                            static_cast<unsigned>(end_bit), stream));
  *sorted_in_b = keys.current() == d_b;
  return AG_OK;
}

size_t ag_scan_i32_temp_bytes(int64_t n) {
  size_t bytes = 0;
  int* p = nullptr;
  (void)rocprim::exclusive_scan(nullptr, bytes, p, p, 0, static_cast<size_t>(n > 0 ? n : 1), rocprim::plus<int>(),
                                hipStream_t(0));
  return bytes + 256;
}

int ag_exclusive_scan_i32(void* d_temp, size_t temp_bytes, const int* d_in, int* d_out, int64_t n, hipStream_t stream) {
  if (n <= 0) return AG_OK;
  AgProfScope prof("rocprim_exclusive_scan", stream);
  AG_HIP(rocprim::exclusive_scan(d_temp, temp_bytes, d_in, d_out, 0, static_cast<size_t>(n), rocprim::plus<int>(),
                                 stream));
  return AG_OK;
}

__global__ void ag_zero_i32_kernel(int* __restrict__ p, int64_t n) {
  for (int64_t i = blockIdx.x * int64_t(blockDim.x) + threadIdx.x; i < n; i += int64_t(gridDim.x) * blockDim.x) p[i] = 0;
}

int ag_zero_i32(int* d_ptr, int64_t n, hipStream_t stream) {
  if (n <= 0) return AG_OK;
  AG_KERNEL("ag_zero_i32_kernel", ag_zero_i32_kernel, dim3(ag_blocks(n, 1024, 4096)), dim3(256), stream, d_ptr, n);
  AG_LAUNCH_CHECK();
  return AG_OK;
}

// ---------------------------------------------------------------------------------------------
// The library's status word: one per device, set by a kernel that had to give up (a scan whose state words were
// not clear) and by the kernels that found such a result and left safe, empty output in its place.  The host
// reads and clears it where it synchronises anyway (ag_status_take).
// ---------------------------------------------------------------------------------------------
static __device__ int ag_status_word_storage[64];  // (a cache line of its own)
#define AG_STATUS_MAX_DEVICES 64

int* ag_status_word() {
  static std::mutex mutex;
  static int* address[AG_STATUS_MAX_DEVICES] = {};
  int device = 0;
  if (hipGetDevice(&device) != hipSuccess || device < 0 || device >= AG_STATUS_MAX_DEVICES) return nullptr;
  std::lock_guard<std::mutex> lock(mutex);
  if (address[device] == nullptr) {
    void* p = nullptr;
    if (hipGetSymbolAddress(&p, HIP_SYMBOL(ag_status_word_storage)) != hipSuccess) return nullptr;
    address[device] = static_cast<int*>(p);  // (device memory of a code object starts out zeroed)
  }
  return address[device];
}

static bool ag_scans_poisoned = false;
bool ag_debug_scans_poisoned() { return ag_scans_poisoned; }
extern "C" int ag_debug_poison_scans(int on) {
  ag_scans_poisoned = on != 0;
  return AG_OK;
}

extern "C" int ag_status_take(int* status) {
  AG_REQUIRE(status != nullptr, "ag_status_take: null pointer");
  int* word = ag_status_word();
  if (word == nullptr) return ag_fail(AG_ERR_HIP, "ag_status_take: the status word of the current device is not available");
  int value = 0;
  AG_HIP(hipMemcpy(&value, word, sizeof(int), hipMemcpyDeviceToHost));  // (waits for the work queued before it)
  if (value != 0) {
    const int zero = 0;
    AG_HIP(hipMemcpy(word, &zero, sizeof(int), hipMemcpyHostToDevice));
  }
  *status = value;
  return AG_OK;
}

// ---------------------------------------------------------------------------------------------
// Single-launch exclusive scan (chained tiles with decoupled look-back).
//
// Tiles take a ticket when they start, so a tile only ever waits for tiles that are already
// running, and those wait for nothing but their own data: every wait ends.  A tile publishes its
// aggregate, then looks back over its predecessors 64 at a time (one per lane of its first wave)
// until it meets an inclusive prefix, and publishes its own inclusive prefix.  One 64-bit word per
// tile carries flag (top two bits: 0 empty, 1 aggregate, 2 inclusive prefix) and value together, so
// a reader can never see one without the other.
//
// The state words -- AG_SCAN_HEADER + number of tiles -- must be ZERO when the kernel starts; the
// callers have a kernel upstream that clears them along with its own counters, which is what makes
// this one launch instead of an initialisation launch plus a scan launch.
// out[n] receives the total (and *total_out, if given); with want_max, out[n + 1] the largest input
// value.  in == out is allowed.
// ---------------------------------------------------------------------------------------------
#define AG_SCAN_HEADER 8  // [0] ticket, [1] tiles finished, [2] running maximum, [3] error word
// A tile that has polled a predecessor this often without seeing it publish gives up: it sets the error word,
// counts the missing predecessors as zero and goes on, so that every block of the launch ends whatever the state
// words held (stale words of a workspace nobody cleared would otherwise be waited for for ever).  The tile that
// writes the total writes -1 instead when the error word is set.  (A poll takes about 0.1 us: the limit is some
// tenths of a second, far beyond any wait for a tile that is actually running.)  Whoever sets the error word also
// sets the library's status word, which the host reports at its next synchronisation.
#define AG_SCAN_SPIN_LIMIT (1 << 22)

template <class T>
struct AgScanTile;
template <>
struct AgScanTile<int> {
  static constexpr int ITEMS = 16;
};
template <>
struct AgScanTile<int64_t> {
  static constexpr int ITEMS = 8;
};

template <class T>
static int64_t ag_scan_tiles(int64_t n) {
  const int64_t tile = 256 * AgScanTile<T>::ITEMS;
  return n > 0 ? (n + tile - 1) / tile : 1;
}

size_t ag_chained_scan_state_words_i32(int64_t n) { return AG_SCAN_HEADER + ag_scan_tiles<int>(n); }
size_t ag_chained_scan_state_words_i64(int64_t n) { return AG_SCAN_HEADER + ag_scan_tiles<int64_t>(n); }

// Tile states are written with an atomic exchange and read with device-scope atomic loads; a reader that
// has polled many times without seeing a value switches to a read-modify-write atomic, which is
// performed where every compute die of the device sees the same value whatever a cache may hold.
__device__ __forceinline__ unsigned long long ag_scan_state_load(unsigned long long* p, int polls) {
  if (polls < 256) return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  return __hip_atomic_fetch_or(p, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void ag_scan_state_store(unsigned long long* p, unsigned long long v) {
  (void)__hip_atomic_exchange(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

template <class T>
__global__ __launch_bounds__(256) void ag_chained_scan_kernel(const T* in, T* out, int64_t n,
                                                              unsigned long long* __restrict__ state, int want_max,
                                                              long long* __restrict__ total_out, int spin_limit,
                                                              int* __restrict__ status) {
  constexpr int ITEMS = AgScanTile<T>::ITEMS;
  constexpr unsigned long long FLAG_AGG = 1ull << 62, FLAG_PREFIX = 2ull << 62, VALUE = (1ull << 62) - 1ull;
  __shared__ unsigned s_tile;
  __shared__ long long s_wave[4];
  __shared__ long long s_prefix;
  __shared__ long long s_max[4];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if (tid == 0) s_tile = atomicAdd(reinterpret_cast<unsigned*>(state), 1u);
  __syncthreads();
  const int64_t tile = s_tile;
  const int64_t ntiles = (n + 256 * ITEMS - 1) / (256 * ITEMS);
  if (tile >= ntiles) {  // a ticket beyond the tiles of this scan: the state words were not zero at launch
    if (tid == 0) {
      ag_scan_state_store(state + 3, 1ull);
      if (status != nullptr) *status = AG_STATUS_SCAN_GAVE_UP;
    }
    return;
  }
  const int64_t first = tile * (256 * ITEMS) + int64_t(tid) * ITEMS;
  long long v[ITEMS];
  long long sum = 0, mx = 0;
  // A thread's items are 64 consecutive bytes: fetched (and stored, below) as four 16-byte pieces wherever the tile
  // is whole and the arrays are aligned.  Item by item, every load instruction of a wave touched 64 lines for 4 bytes
  // each, and the scan ran at the rate the L1 takes lines: 0.85 TB/s over 7e7 cell counts.
  typedef int AgScanPiece __attribute__((ext_vector_type(4)));
  static_assert(ITEMS * sizeof(T) == 64, "a thread's items are four 16-byte pieces");
  const bool pieces = (tile + 1) * (256 * ITEMS) <= n &&
                      ((reinterpret_cast<uintptr_t>(in) | reinterpret_cast<uintptr_t>(out)) & 15u) == 0;
  if (pieces) {
    T raw[ITEMS];
    const AgScanPiece* src = reinterpret_cast<const AgScanPiece*>(in + first);
#pragma unroll
    for (int p = 0; p < 4; ++p) *reinterpret_cast<AgScanPiece*>(raw + p * (ITEMS / 4)) = src[p];
#pragma unroll
    for (int i = 0; i < ITEMS; ++i) v[i] = static_cast<long long>(raw[i]);
  } else {
#pragma unroll
    for (int i = 0; i < ITEMS; ++i) v[i] = (first + i < n) ? static_cast<long long>(in[first + i]) : 0;
  }
#pragma unroll
  for (int i = 0; i < ITEMS; ++i) {
    sum += v[i];
    mx = v[i] > mx ? v[i] : mx;
  }
  // exclusive scan of the per-thread sums inside the block
  long long inc = sum;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const long long up = __shfl_up(inc, o, 64);
    if (lane >= o) inc += up;
  }
  if (lane == 63) s_wave[wave] = inc;
  if (want_max) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      const long long other = __shfl_xor(mx, o, 64);
      mx = other > mx ? other : mx;
    }
    if (lane == 0) s_max[wave] = mx;
  }
  __syncthreads();
  long long wave_base = 0, block_total = 0;
#pragma unroll
  for (int w = 0; w < 4; ++w) {
    if (w < wave) wave_base += s_wave[w];
    block_total += s_wave[w];
  }
  const long long thread_excl = wave_base + inc - sum;

  if (wave == 0) {
    long long prefix = 0;
    if (tile > 0) {
      if (lane == 0) ag_scan_state_store(state + AG_SCAN_HEADER + tile, FLAG_AGG | static_cast<unsigned long long>(block_total));
      int64_t look = tile - 1;
      while (true) {
        const int64_t idx = look - lane;  // lane 0 reads the nearest predecessor
        int polls = 0;
        unsigned long long st = idx >= 0 ? ag_scan_state_load(state + AG_SCAN_HEADER + idx, polls) : FLAG_PREFIX;
        while (__any((st >> 62) == 0ull)) {
          __builtin_amdgcn_s_sleep(1);
          ++polls;
          if (idx >= 0 && (st >> 62) == 0ull) st = ag_scan_state_load(state + AG_SCAN_HEADER + idx, polls);
          if (polls >= spin_limit) {  // (wave-uniform) give up: see AG_SCAN_SPIN_LIMIT
            if ((st >> 62) == 0ull) st = FLAG_PREFIX;  // the missing predecessors count as zero; the look-back ends here
            if (lane == 0) {
              ag_scan_state_store(state + 3, 1ull);
              if (status != nullptr) *status = AG_STATUS_SCAN_GAVE_UP;
            }
          }
        }
        const unsigned long long has_prefix = __ballot((st >> 62) == 2ull);
        const int stop = has_prefix ? (__ffsll(static_cast<long long>(has_prefix)) - 1) : 64;
        long long part = (lane <= stop) ? static_cast<long long>(st & VALUE) : 0;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) part += __shfl_xor(part, o, 64);
        prefix += part;
        if (has_prefix) break;
        look -= 64;
      }
    }
    if (lane == 0) {
      ag_scan_state_store(state + AG_SCAN_HEADER + tile, FLAG_PREFIX | static_cast<unsigned long long>(prefix + block_total));
      s_prefix = prefix;
    }
  }
  __syncthreads();
  long long run = s_prefix + thread_excl;
  if (pieces) {
    T res[ITEMS];
#pragma unroll
    for (int i = 0; i < ITEMS; ++i) {
      res[i] = static_cast<T>(run);
      run += v[i];
    }
    AgScanPiece* dst = reinterpret_cast<AgScanPiece*>(out + first);
#pragma unroll
    for (int p = 0; p < 4; ++p) dst[p] = *reinterpret_cast<const AgScanPiece*>(res + p * (ITEMS / 4));
  } else {
#pragma unroll
    for (int i = 0; i < ITEMS; ++i) {
      if (first + i < n) out[first + i] = static_cast<T>(run);
      run += v[i];
    }
  }
  if (tile == ntiles - 1 && tid == 255) {
    // (every predecessor has published before this tile got here, so an error any of them met is visible)
    const bool failed = __hip_atomic_load(state + 3, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0ull;
    const long long total = failed ? -1ll : s_prefix + block_total;
    out[n] = static_cast<T>(total);
    if (total_out != nullptr) *total_out = total;
  }
  if (want_max && tid == 0) {
    long long m = s_max[0];
    for (int w = 1; w < 4; ++w) m = s_max[w] > m ? s_max[w] : m;
    atomicMax(state + 2, static_cast<unsigned long long>(m));
    __threadfence();
    const unsigned long long done = atomicAdd(state + 1, 1ull);
    if (done == static_cast<unsigned long long>(ntiles - 1)) out[n + 1] = static_cast<T>(atomicMax(state + 2, 0ull));
  }
}

// d_state: ag_chained_scan_state_words_*(n) zeroed 64-bit words (see above)
int ag_chained_scan_i32(const int* d_in, int* d_out, int64_t n, unsigned long long* d_state, hipStream_t stream,
                        long long* d_total) {
  if (n <= 0) return AG_OK;
  AG_KERNEL("ag_chained_scan_kernel", ag_chained_scan_kernel<int>, dim3(static_cast<unsigned>(ag_scan_tiles<int>(n))),
            dim3(256), stream, d_in, d_out, n, d_state, 0, d_total, AG_SCAN_SPIN_LIMIT, ag_status_word());
  AG_LAUNCH_CHECK();
  return AG_OK;
}

int ag_chained_scan_i64(const int64_t* d_in, int64_t* d_out, int64_t n, unsigned long long* d_state, int want_max,
                        hipStream_t stream) {
  if (n <= 0) return AG_OK;
  AG_KERNEL("ag_chained_scan_kernel", ag_chained_scan_kernel<int64_t>,
            dim3(static_cast<unsigned>(ag_scan_tiles<int64_t>(n))), dim3(256), stream, d_in, d_out, n, d_state, want_max,
            static_cast<long long*>(nullptr), AG_SCAN_SPIN_LIMIT, ag_status_word());
  AG_LAUNCH_CHECK();
  return AG_OK;
}

// The single-launch scan on its own, with the bound of its look-back wait as an argument (tests force the
// give-up path with stale state words and a tiny bound; every block still ends).  d_state: *d_state_words 64-bit
// words, all zero for a valid scan; d_out: n + 1 entries, d_out[n] = total, or -1 if a tile gave up.
extern "C" size_t ag_scan_i32_state_words(int64_t n) { return ag_chained_scan_state_words_i32(n); }
extern "C" int ag_scan_i32(const int32_t* d_in, int32_t* d_out, int64_t n, unsigned long long* d_state, int spin_limit,
                           void* stream) {
  AG_REQUIRE(n >= 1 && n < (int64_t(1) << 31), "ag_scan_i32: size out of range");
  AG_REQUIRE(spin_limit >= 1, "ag_scan_i32: the spin limit must be positive");
  AG_REQUIRE(d_in && d_out && d_state, "ag_scan_i32: null pointer");
  hipStream_t s = static_cast<hipStream_t>(stream);
  AG_KERNEL("ag_chained_scan_kernel", ag_chained_scan_kernel<int>, dim3(static_cast<unsigned>(ag_scan_tiles<int>(n))),
            dim3(256), s, d_in, d_out, n, d_state, 0, static_cast<long long*>(nullptr), spin_limit, ag_status_word());
  AG_LAUNCH_CHECK();
  return AG_OK;
}

size_t ag_scan_i64_temp_bytes(int64_t n) {
  size_t bytes = 0;
  int64_t* p = nullptr;
  (void)rocprim::exclusive_scan(nullptr, bytes, p, p, int64_t(0), static_cast<size_t>(n + 1), rocprim::plus<int64_t>(),
                                hipStream_t(0));
  return bytes + 256;
}

__global__ void ag_zero_i64_kernel(int64_t* p) { *p = 0; }

int ag_exclusive_scan_i64(void* d_temp, size_t temp_bytes, const int64_t* d_in, int64_t* d_out, int64_t n,
                          hipStream_t stream) {
  // scan n+1 entries with a zero appended so that d_out[n] holds the total
  int64_t* in = const_cast<int64_t*>(d_in);
  AG_KERNEL("ag_zero_i64_kernel", ag_zero_i64_kernel, dim3(1), dim3(1), stream, in + n);
  AG_LAUNCH_CHECK();
  AgProfScope prof("rocprim_exclusive_scan", stream);
  AG_HIP(rocprim::exclusive_scan(d_temp, temp_bytes, in, d_out, int64_t(0), static_cast<size_t>(n + 1),
                                 rocprim::plus<int64_t>(), stream));
  return AG_OK;
}

// ---------------------------------------------------------------------------------------------
// masking path
// ---------------------------------------------------------------------------------------------
extern "C" size_t ag_mask_compact_workspace_bytes(int64_t n) {
  size_t bytes = 0;
  rocprim::counting_iterator<int32_t> ids(0);
  (void)rocprim::select(nullptr, bytes, ids, static_cast<const uint8_t*>(nullptr), static_cast<int32_t*>(nullptr),
                        static_cast<int64_t*>(nullptr), static_cast<size_t>(n > 0 ? n : 1), hipStream_t(0));
  return bytes + 512;
}

extern "C" int ag_mask_compact(const uint8_t* d_mask, int64_t n, int32_t* d_ids, int64_t* d_count, void* d_ws,
                               size_t ws_bytes, void* stream) {
  AG_REQUIRE(n >= 0 && n < (int64_t(1) << 31), "ag_mask_compact: n out of range");
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (n == 0) {
    AG_KERNEL("ag_zero_i64_kernel", ag_zero_i64_kernel, dim3(1), dim3(1), s, d_count);
    AG_LAUNCH_CHECK();
    return AG_OK;
  }
  if (ws_bytes < ag_mask_compact_workspace_bytes(n)) return ag_fail(AG_ERR_WORKSPACE, "ag_mask_compact: workspace too small");
  rocprim::counting_iterator<int32_t> ids(0);
  size_t bytes = ws_bytes;
  AG_HIP(rocprim::select(d_ws, bytes, ids, d_mask, d_ids, d_count, static_cast<size_t>(n), s));
  return AG_OK;
}

// Ids that index a table are checked against its length in every kernel below: an id outside [0, length) is
// never followed (the entry is skipped, or filled with a fixed value), and the caller's error word *bad is
// set to 1 -- every writer stores the same value.  numpy / torch raise IndexError for such an id in the
// reference; the host side of this library does so when it reads the word.
__global__ void ag_gather_coords_kernel(const float2* __restrict__ x, int64_t n_x, const int32_t* __restrict__ ids,
                                        int64_t n, float2* __restrict__ out, int* __restrict__ bad) {
  for (int64_t i = blockIdx.x * int64_t(blockDim.x) + threadIdx.x; i < n; i += int64_t(gridDim.x) * blockDim.x) {
    const int32_t id = ids[i];
    const bool ok = id >= 0 && id < n_x;
    out[i] = ok ? x[id] : make_float2(0.0f, 0.0f);
    if (!ok && bad != nullptr) *bad = 1;
  }
}

extern "C" int ag_gather_coords(const float* d_x, int64_t n_x, const int32_t* d_ids, int64_t n, float* d_out,
                                int32_t* d_bad, void* stream) {
  AG_REQUIRE(n >= 0 && n_x >= 0, "ag_gather_coords: negative size");
  if (n == 0) return AG_OK;
  AG_REQUIRE(d_ids && d_out && (d_x || n_x == 0), "ag_gather_coords: null pointer");
  AG_KERNEL("ag_gather_coords_kernel", ag_gather_coords_kernel, dim3(ag_blocks(n, 256, 8192)), dim3(256), static_cast<hipStream_t>(stream), reinterpret_cast<const float2*>(d_x), n_x, d_ids, n, reinterpret_cast<float2*>(d_out), d_bad);
  AG_LAUNCH_CHECK();
  return AG_OK;
}

__global__ void ag_remap_ids_kernel(int32_t* __restrict__ v, int64_t n, const int32_t* __restrict__ map, int64_t n_map,
                                    int* __restrict__ bad) {
  for (int64_t i = blockIdx.x * int64_t(blockDim.x) + threadIdx.x; i < n; i += int64_t(gridDim.x) * blockDim.x) {
    const int32_t id = v[i];
    const bool ok = id >= 0 && id < n_map;
    v[i] = ok ? map[id] : -1;
    if (!ok && bad != nullptr) *bad = 1;
  }
}

extern "C" int ag_remap_ids(int32_t* d_values, int64_t n, const int32_t* d_map, int64_t n_map, int32_t* d_bad,
                            void* stream) {
  AG_REQUIRE(n >= 0 && n_map >= 0, "ag_remap_ids: negative size");
  if (n == 0) return AG_OK;
  AG_REQUIRE(d_values && (d_map || n_map == 0), "ag_remap_ids: null pointer");
  AG_KERNEL("ag_remap_ids_kernel", ag_remap_ids_kernel, dim3(ag_blocks(n, 256, 8192)), dim3(256), static_cast<hipStream_t>(stream), d_values, n, d_map, n_map, d_bad);
  AG_LAUNCH_CHECK();
  return AG_OK;
}

// ---------------------------------------------------------------------------------------------
// edge list sort / unique
// ---------------------------------------------------------------------------------------------
__global__ void ag_pack_edges_kernel(const int32_t* __restrict__ src, const int32_t* __restrict__ tgt, int64_t E,
                                     int by_target, uint64_t* __restrict__ keys) {
  for (int64_t i = blockIdx.x * int64_t(blockDim.x) + threadIdx.x; i < E; i += int64_t(gridDim.x) * blockDim.x) {
    // ids may be negative (the reference's own concat_edges test uses negative ids): flip the sign
    // bit so that the unsigned key order equals the signed order
    const uint64_t s = static_cast<uint32_t>(src[i]) ^ 0x80000000u, t = static_cast<uint32_t>(tgt[i]) ^ 0x80000000u;
    keys[i] = by_target ? ((t << 32) | s) : ((s << 32) | t);
  }
}

__global__ void ag_unpack_edges_kernel(const uint64_t* __restrict__ keys, int64_t E, int by_target,
                                       int32_t* __restrict__ src, int32_t* __restrict__ tgt) {
  for (int64_t i = blockIdx.x * int64_t(blockDim.x) + threadIdx.x; i < E; i += int64_t(gridDim.x) * blockDim.x) {
    const uint64_t k = keys[i];
    const int32_t hi = static_cast<int32_t>(static_cast<uint32_t>(k >> 32) ^ 0x80000000u);
    const int32_t lo = static_cast<int32_t>(static_cast<uint32_t>(k) ^ 0x80000000u);
    src[i] = by_target ? lo : hi;
    tgt[i] = by_target ? hi : lo;
  }
}

__global__ void ag_unpack_unique_kernel(const uint64_t* __restrict__ keys, const int64_t* __restrict__ count,
                                        int32_t* __restrict__ src, int32_t* __restrict__ tgt) {
  const int64_t E = *count;
  for (int64_t i = blockIdx.x * int64_t(blockDim.x) + threadIdx.x; i < E; i += int64_t(gridDim.x) * blockDim.x) {
    const uint64_t k = keys[i];
    src[i] = static_cast<int32_t>(static_cast<uint32_t>(k >> 32) ^ 0x80000000u);
    tgt[i] = static_cast<int32_t>(static_cast<uint32_t>(k) ^ 0x80000000u);
  }
}

static size_t ag_unique_temp_bytes(int64_t E) {
  size_t bytes = 0;
  uint64_t* p = nullptr;
  (void)rocprim::unique(nullptr, bytes, p, p, static_cast<int64_t*>(nullptr), static_cast<size_t>(E > 0 ? E : 1),
                        rocprim::equal_to<uint64_t>(), hipStream_t(0));
  return bytes + 256;
}

struct AgSortWs {
  uint64_t *a, *b;
  void* temp;
  size_t temp_bytes;
};

static size_t ag_sort_ws_carve(AgCarver& c, int64_t E, AgSortWs* w) {
  const int64_t n = E > 0 ? E : 1;
  size_t t1 = ag_sort_u64_temp_bytes(n), t2 = ag_unique_temp_bytes(n);
  size_t tb = t1 > t2 ? t1 : t2;
  uint64_t* a = c.take<uint64_t>(n);
  uint64_t* b = c.take<uint64_t>(n);
  void* temp = c.take<char>(tb);
  if (w) *w = {a, b, temp, tb};
  return c.total();
}

extern "C" size_t ag_sort_edges_workspace_bytes(int64_t E) {
  AgCarver c(nullptr);
  return ag_sort_ws_carve(c, E, nullptr);
}

extern "C" int ag_sort_edges(int32_t* d_src, int32_t* d_tgt, int64_t E, int by_target, void* d_ws, size_t ws_bytes,
                             void* stream) {
  AG_REQUIRE(E >= 0, "ag_sort_edges: negative edge count");
  if (E == 0) return AG_OK;
  if (ws_bytes < ag_sort_edges_workspace_bytes(E)) return ag_fail(AG_ERR_WORKSPACE, "ag_sort_edges: workspace too small");
  hipStream_t s = static_cast<hipStream_t>(stream);
  AgCarver c(d_ws);
  AgSortWs w;
  ag_sort_ws_carve(c, E, &w);
  const int nb = ag_blocks(E, 256, 8192);
  AG_KERNEL("ag_pack_edges_kernel", ag_pack_edges_kernel, dim3(nb), dim3(256), s, d_src, d_tgt, E, by_target, w.a);
  AG_LAUNCH_CHECK();
  int rc = ag_sort_u64(w.temp, w.temp_bytes, w.a, w.b, E, 0, 64, s);
  if (rc) return rc;
  AG_KERNEL("ag_unpack_edges_kernel", ag_unpack_edges_kernel, dim3(nb), dim3(256), s, w.b, E, by_target, d_src, d_tgt);
  AG_LAUNCH_CHECK();
  return AG_OK;
}

extern "C" int ag_unique_edges(int32_t* d_src, int32_t* d_tgt, int64_t E, int64_t* d_num_unique, void* d_ws,
                               size_t ws_bytes, void* stream) {
  AG_REQUIRE(E >= 0, "ag_unique_edges: negative edge count");
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (E == 0) {
    AG_KERNEL("ag_zero_i64_kernel", ag_zero_i64_kernel, dim3(1), dim3(1), s, d_num_unique);
    AG_LAUNCH_CHECK();
    return AG_OK;
  }
  if (ws_bytes < ag_sort_edges_workspace_bytes(E)) return ag_fail(AG_ERR_WORKSPACE, "ag_unique_edges: workspace too small");
  AgCarver c(d_ws);
  AgSortWs w;
  ag_sort_ws_carve(c, E, &w);
  const int nb = ag_blocks(E, 256, 8192);
  AG_KERNEL("ag_pack_edges_kernel", ag_pack_edges_kernel, dim3(nb), dim3(256), s, d_src, d_tgt, E, 0, w.a);
  AG_LAUNCH_CHECK();
  int rc = ag_sort_u64(w.temp, w.temp_bytes, w.a, w.b, E, 0, 64, s);
  if (rc) return rc;
  size_t tb = w.temp_bytes;
  AG_HIP(rocprim::unique(w.temp, tb, w.b, w.a, d_num_unique, static_cast<size_t>(E), rocprim::equal_to<uint64_t>(), s));
  AG_KERNEL("ag_unpack_unique_kernel", ag_unpack_unique_kernel, dim3(nb), dim3(256), s, w.a, d_num_unique, d_src, d_tgt);
  AG_LAUNCH_CHECK();
  return AG_OK;
}

// ---------------------------------------------------------------------------------------------
// target column of a fixed-degree edge list: out[r * k + j] = r
// ---------------------------------------------------------------------------------------------
__global__ void ag_fill_row_ids_kernel(int32_t* __restrict__ out, int64_t n, int k) {
  for (int64_t i = blockIdx.x * int64_t(blockDim.x) + threadIdx.x; i < n; i += int64_t(gridDim.x) * blockDim.x)
    out[i] = static_cast<int32_t>(i / k);
}

extern "C" int ag_fill_row_ids(int32_t* d_out, int64_t rows, int k, void* stream) {
  AG_REQUIRE(rows >= 0 && k >= 1, "ag_fill_row_ids: bad sizes");
  AG_REQUIRE(rows < (int64_t(1) << 31), "ag_fill_row_ids: row ids must fit int32");
  if (rows == 0) return AG_OK;
  AG_REQUIRE(d_out != nullptr, "ag_fill_row_ids: null pointer");
  AG_KERNEL("ag_fill_row_ids_kernel", ag_fill_row_ids_kernel, dim3(ag_blocks(rows * k, 1024, 8192)), dim3(256),
            static_cast<hipStream_t>(stream), d_out, rows * k, k);
  AG_LAUNCH_CHECK();
  return AG_OK;
}

// ---------------------------------------------------------------------------------------------
// connected-node masks and index compaction maps (processors/post_process.py:48-61,133-149)
// ---------------------------------------------------------------------------------------------
__global__ void ag_mark_ids_kernel(const int32_t* __restrict__ ids, int64_t n, uint8_t* __restrict__ mask, int64_t n_mask,
                                   int* __restrict__ bad) {
  for (int64_t i = blockIdx.x * int64_t(blockDim.x) + threadIdx.x; i < n; i += int64_t(gridDim.x) * blockDim.x) {
    const int32_t v = ids[i];
    if (v >= 0 && v < n_mask) mask[v] = 1;  // every writer stores the same byte
    else if (bad != nullptr) *bad = 1;
  }
}

extern "C" int ag_mark_ids(const int32_t* d_ids, int64_t n, uint8_t* d_mask, int64_t n_mask, int32_t* d_bad,
                           void* stream) {
  AG_REQUIRE(n >= 0 && n_mask >= 0, "ag_mark_ids: negative size");
  if (n == 0) return AG_OK;
  AG_REQUIRE(d_ids && (d_mask || n_mask == 0), "ag_mark_ids: null pointer");
  AG_KERNEL("ag_mark_ids_kernel", ag_mark_ids_kernel, dim3(ag_blocks(n, 256, 8192)), dim3(256),
            static_cast<hipStream_t>(stream), d_ids, n, d_mask, n_mask, d_bad);
  AG_LAUNCH_CHECK();
  return AG_OK;
}

__global__ void ag_invert_ids_kernel(const int32_t* __restrict__ ids, int64_t n, int32_t* __restrict__ map,
                                     int64_t n_map, int* __restrict__ bad) {
  for (int64_t i = blockIdx.x * int64_t(blockDim.x) + threadIdx.x; i < n; i += int64_t(gridDim.x) * blockDim.x) {
    const int32_t id = ids[i];
    if (id >= 0 && id < n_map) map[id] = static_cast<int32_t>(i);
    else if (bad != nullptr) *bad = 1;
  }
}

extern "C" int ag_invert_ids(const int32_t* d_ids, int64_t n, int32_t* d_map, int64_t n_map, int32_t* d_bad,
                             void* stream) {
  AG_REQUIRE(n >= 0 && n_map >= 0, "ag_invert_ids: negative size");
  AG_REQUIRE(n < (int64_t(1) << 31), "ag_invert_ids: positions must fit int32");
  if (n == 0) return AG_OK;
  AG_REQUIRE(d_ids && (d_map || n_map == 0), "ag_invert_ids: null pointer");
  AG_KERNEL("ag_invert_ids_kernel", ag_invert_ids_kernel, dim3(ag_blocks(n, 256, 8192)), dim3(256),
            static_cast<hipStream_t>(stream), d_ids, n, d_map, n_map, d_bad);
  AG_LAUNCH_CHECK();
  return AG_OK;
}
