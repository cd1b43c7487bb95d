// rbx_route.hip -- C1 routing for row-sharded tables: wire-slot assignment of the padded,
// sync-free all-to-all (gfx950).
//
// The reference has no model-parallel embedding; this is the build's own
// exchange format (recbox_amd/sharded.py):  lookup i = (sample b, table t) with id x goes to rank
// owner = x mod W, local row base[owner][t] + x div W, and occupies wire slot
//     owner * capacity + (number of earlier lookups with the same owner)
// -- a stable counting sort by owner, so slot order == lookup order inside every owner block and the
// result is deterministic.  Lookups beyond an owner's capacity get slot W*capacity ("dump": read as
// a zero row, no gradient) and raise the overflow byte instead of being dropped silently.
//
// Three launches, no host sync: per-tile owner histogram -> exclusive scan over tiles (one workgroup
// per owner) -> rank assignment (ballot + popcount inside a wavefront, LDS across the 4 wavefronts
// and the 8 rounds of a tile).  HBM traffic: ids twice (8 B each) + slot (4 B) + send (8 B).
#include "rbx_internal.h"

namespace rbx {

constexpr int kRouteTile = 2048;      // lookups per workgroup: 8 rounds of 256
constexpr int kRouteMaxW = 64;

struct RouteField {          // ids of one table: a strided, typed column read in place
  const void* ids;
  long long stride_b;
  int dtype;
  int pad;
};
struct RoutePack { RouteField f[RBX_MAX_FIELDS]; };

// lookup i = (sample i / T, table i % T)
__device__ __forceinline__ long long route_id(const RoutePack& P, long long i, int T) {
  const long long b = i / T;
  const RouteField& fd = P.f[static_cast<int>(i - b * T)];
  return load_id(fd.ids, b * fd.stride_b, fd.dtype);
}

__device__ __forceinline__ int owner_of(long long id, int W) {
  int o = static_cast<int>(id % W);
  return o < 0 ? o + W : o;
}

__global__ __launch_bounds__(256) void route_count_kernel(const RoutePack P, const int T, const long long n,
                                                          const int W, int* __restrict__ hist /*[W][tiles]*/,
                                                          const int tiles) {
  __shared__ int s_cnt[kRouteMaxW];
  if (threadIdx.x < W) s_cnt[threadIdx.x] = 0;
  __syncthreads();
  const long long first = static_cast<long long>(blockIdx.x) * kRouteTile;
#pragma unroll
  for (int j = 0; j < kRouteTile / 256; ++j) {
    const long long i = first + j * 256 + threadIdx.x;
    if (i < n) atomicAdd(&s_cnt[owner_of(route_id(P, i, T), W)], 1);     // integer LDS atomics: order-independent
  }
  __syncthreads();
  if (threadIdx.x < W) hist[static_cast<long long>(threadIdx.x) * tiles + blockIdx.x] = s_cnt[threadIdx.x];
}

// one workgroup per owner: exclusive scan of its tile counts (in place), total vs capacity
__global__ __launch_bounds__(256) void route_scan_kernel(int* __restrict__ hist, const int tiles, const long long capacity,
                                                         unsigned char* __restrict__ overflow) {
  __shared__ int s_wave[4];
  __shared__ int s_carry;
  int* col = hist + static_cast<long long>(blockIdx.x) * tiles;
  if (threadIdx.x == 0) s_carry = 0;
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int t0 = 0; t0 < tiles; t0 += 256) {
    const int t = t0 + threadIdx.x;
    const int v = (t < tiles) ? col[t] : 0;
    int incl = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const int up = __shfl_up(incl, o, 64);
      if (lane >= o) incl += up;
    }
    if (lane == 63) s_wave[wave] = incl;
    __syncthreads();
    int before = s_carry;
    for (int w = 0; w < wave; ++w) before += s_wave[w];
    if (t < tiles) col[t] = before + incl - v;
    __syncthreads();
    if (threadIdx.x == 255) s_carry = before + incl;
    __syncthreads();
  }
  if (threadIdx.x == 0 && s_carry > capacity && overflow != nullptr) *overflow = 1;
}

template <typename WireT>
__global__ __launch_bounds__(256) void route_assign_kernel(const RoutePack P, const long long n,
                                                           const int T, const int W, const long long capacity,
                                                           const long long* __restrict__ base /*[W][T]*/,
                                                           const int* __restrict__ hist, const int tiles,
                                                           WireT* __restrict__ send, int* __restrict__ slot) {
  __shared__ int s_run[kRouteMaxW];            // owner's lookups before the current round
  __shared__ int s_wave[4][kRouteMaxW];        // per wavefront counts of the current round
  if (threadIdx.x < W) s_run[threadIdx.x] = hist[static_cast<long long>(threadIdx.x) * tiles + blockIdx.x];
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const unsigned long long below = (1ull << lane) - 1ull;
  const long long first = static_cast<long long>(blockIdx.x) * kRouteTile;
  const long long dump = capacity * W;
  for (int j = 0; j < kRouteTile / 256; ++j) {
    const long long i = first + j * 256 + threadIdx.x;
    const bool valid = i < n;
    const long long id = valid ? route_id(P, i, T) : 0;
    const int own = valid ? owner_of(id, W) : -1;
    int in_wave = 0;
    for (int w = 0; w < W; ++w) {              // W is small (GPUs of one node)
      const unsigned long long m = __ballot(own == w);
      if (own == w) in_wave = __popcll(m & below);
      if (lane == 0) s_wave[wave][w] = __popcll(m);
    }
    __syncthreads();
    if (valid) {
      int rank = s_run[own] + in_wave;
      for (int v = 0; v < wave; ++v) rank += s_wave[v][own];
      const int t = static_cast<int>(i % T);
      const long long row = base[static_cast<long long>(own) * T + t] + (id - own) / W;
      if (rank < capacity) {
        const long long s = static_cast<long long>(own) * capacity + rank;
        slot[i] = static_cast<int>(s);
        send[s] = static_cast<WireT>(row);
      } else {
        slot[i] = static_cast<int>(dump);
      }
    }
    __syncthreads();
    if (threadIdx.x < W) s_run[threadIdx.x] += s_wave[0][threadIdx.x] + s_wave[1][threadIdx.x] + s_wave[2][threadIdx.x] +
                                               s_wave[3][threadIdx.x];
    __syncthreads();
  }
}

static long long route_tiles(long long n) { return (n + kRouteTile - 1) / kRouteTile; }

}  // namespace rbx

extern "C" size_t rbx_route_workspace_size(int64_t n_lookups, int32_t world) {
  if (n_lookups <= 0 || world <= 0) return 0;
  return static_cast<size_t>(rbx::route_tiles(n_lookups)) * static_cast<size_t>(world) * sizeof(int);
}

namespace rbx {
template <typename WireT>
static int route_impl(const rbx_field_t* tables, int32_t n_tables, int64_t batch, int32_t world, int64_t capacity,
                      const int64_t* d_base, WireT* d_send, int32_t* d_slot, uint8_t* d_overflow, void* d_workspace,
                      size_t workspace_bytes, void* stream) {
  if (batch < 0 || n_tables <= 0 || n_tables > RBX_MAX_FIELDS || tables == nullptr)
    return fail(RBX_ERR_INVALID, "route: bad sizes (batch %lld, %d tables)", static_cast<long long>(batch), n_tables);
  if (world <= 0 || world > kRouteMaxW) return fail(RBX_ERR_UNSUPPORTED, "route: world=%d not in [1,%d]", world, kRouteMaxW);
  if (capacity <= 0 || capacity * world >= INT_MAX) return fail(RBX_ERR_INVALID, "route: bad capacity");
  hipStream_t s = as_stream(stream);
  if (d_send == nullptr || d_base == nullptr || (batch > 0 && d_slot == nullptr))
    return fail(RBX_ERR_INVALID, "route: NULL output/base");
  RoutePack pack;
  for (int t = 0; t < n_tables; ++t) {
    if (batch > 0 && tables[t].ids == nullptr) return fail(RBX_ERR_INVALID, "route: table %d: ids is NULL", t);
    if (tables[t].ids_dtype < RBX_I32 || tables[t].ids_dtype > RBX_F64) return fail(RBX_ERR_INVALID, "route: table %d: bad ids_dtype", t);
    pack.f[t].ids = tables[t].ids;
    pack.f[t].stride_b = tables[t].ids_stride_b;
    pack.f[t].dtype = tables[t].ids_dtype;
    pack.f[t].pad = 0;
  }
  // empty wire slots carry row -1 (the owner's gather returns a zero row for them)
  if (hipMemsetAsync(d_send, 0xFF, static_cast<size_t>(capacity) * world * sizeof(WireT), s) != hipSuccess)
    return fail(RBX_ERR_LAUNCH, "route: memset failed");
  const long long n_lookups = static_cast<long long>(batch) * n_tables;
  if (n_lookups == 0) return RBX_OK;
  const long long tiles = route_tiles(n_lookups);
  if (tiles >= INT_MAX) return fail(RBX_ERR_UNSUPPORTED, "route: too many lookups");
  if (d_workspace == nullptr || workspace_bytes < rbx_route_workspace_size(n_lookups, world))
    return fail(RBX_ERR_WORKSPACE, "route: workspace too small");
  int* hist = static_cast<int*>(d_workspace);
  hipLaunchKernelGGL(route_count_kernel, dim3(static_cast<unsigned>(tiles)), dim3(256), 0, s, pack, n_tables, n_lookups,
                     world, hist, static_cast<int>(tiles));
  hipLaunchKernelGGL(route_scan_kernel, dim3(world), dim3(256), 0, s, hist, static_cast<int>(tiles),
                     static_cast<long long>(capacity), d_overflow);
  hipLaunchKernelGGL(route_assign_kernel<WireT>, dim3(static_cast<unsigned>(tiles)), dim3(256), 0, s, pack, n_lookups,
                     n_tables, world, static_cast<long long>(capacity), reinterpret_cast<const long long*>(d_base), hist,
                     static_cast<int>(tiles), d_send, d_slot);
  return check_launch("route kernels");
}
}  // namespace rbx

extern "C" int rbx_route(const rbx_field_t* tables, int32_t n_tables, int64_t batch, int32_t world, int64_t capacity,
                         const int64_t* d_base, int64_t* d_send, int32_t* d_slot, uint8_t* d_overflow,
                         void* d_workspace, size_t workspace_bytes, void* stream) {
  return rbx::route_impl<long long>(tables, n_tables, batch, world, capacity, d_base, reinterpret_cast<long long*>(d_send),
                                    d_slot, d_overflow, d_workspace, workspace_bytes, stream);
}

extern "C" int rbx_route32(const rbx_field_t* tables, int32_t n_tables, int64_t batch, int32_t world, int64_t capacity,
                           const int64_t* d_base, int32_t* d_send, int32_t* d_slot, uint8_t* d_overflow,
                           void* d_workspace, size_t workspace_bytes, void* stream) {
  return rbx::route_impl<int>(tables, n_tables, batch, world, capacity, d_base, d_send, d_slot, d_overflow, d_workspace,
                              workspace_bytes, stream);
}
