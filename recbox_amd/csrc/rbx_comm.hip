// rbx_comm.hip -- the exchange step of the row-sharded tables issued straight on the caller's HIP stream.
//
// torch.distributed runs every collective on RCCL's own stream and joins it to the caller's stream with two events;
// between short dependent pieces (route -> ids out -> gather -> rows back -> forward ...) each of those joins costs
// 15-50 us of idle GPU.  Here the all-to-all is the grouped ncclSend / ncclRecv sequence itself, enqueued on the
// stream the neighbouring kernels run on, through the communicator torch.distributed already built (its handle is
// passed in; the RCCL entry points come from the library the process has loaded -- rbx_comm_bind -- so this file
// links against nothing).
#include "rbx_internal.h"

namespace rbx {
typedef int (*nccl_group_fn)();
typedef int (*nccl_p2p_fn)(void* buf, size_t count, int datatype, int peer, void* comm, hipStream_t stream);
typedef const char* (*nccl_err_fn)(int);
typedef int (*nccl_allreduce_fn)(const void* send, void* recv, size_t count, int datatype, int op, void* comm, hipStream_t stream);
typedef int (*nccl_allgather_fn)(const void* send, void* recv, size_t sendcount, int datatype, void* comm, hipStream_t stream);
typedef int (*nccl_rank_fn)(void* comm, int* rank);
static nccl_group_fn g_group_start = nullptr, g_group_end = nullptr;
static nccl_p2p_fn g_send = nullptr, g_recv = nullptr;
static nccl_err_fn g_errstr = nullptr;
static nccl_allreduce_fn g_allreduce = nullptr;
static nccl_allgather_fn g_allgather = nullptr;
static nccl_rank_fn g_userrank = nullptr;
constexpr int kNcclInt8 = 0;      // ncclInt8 / ncclChar
// rbx dtype code (RBX_I32 ... RBX_F64) -> ncclDataType_t (ncclInt32 = 2, ncclInt64 = 4, ncclFloat32 = 7, ncclFloat64 = 8)
static int nccl_type(int dtype) {
  switch (dtype) {
    case RBX_I32: return 2;
    case RBX_I64: return 4;
    case RBX_F32: return 7;
    case RBX_F64: return 8;
    default: return -1;
  }
}
// The block a rank keeps for itself, moved by a kernel of this library: inside a captured step a hipMemcpyAsync becomes a
// memcpy NODE, which the graph runtime runs on a queue of its own -- in the replayed sharded FM step the 46 MB copy of the
// rows' way back started 74 us after the kernel it depends on had finished (profiles/r04/fm_sharded1_replay_timeline.txt);
// a kernel node stays on the queue of its neighbours.
__global__ __launch_bounds__(256) void copy_block_kernel(const uint4* __restrict__ src, uint4* __restrict__ dst, const size_t n16,
                                                         const unsigned char* __restrict__ src_tail,
                                                         unsigned char* __restrict__ dst_tail, const int tail) {
  const size_t stride = static_cast<size_t>(gridDim.x) * blockDim.x;
  for (size_t i = static_cast<size_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < n16; i += stride) dst[i] = src[i];
  if (blockIdx.x == 0 && static_cast<int>(threadIdx.x) < tail) dst_tail[threadIdx.x] = src_tail[threadIdx.x];
}

static int copy_block(const char* src, char* dst, size_t bytes, hipStream_t s) {
  if ((reinterpret_cast<uintptr_t>(src) & 15) != 0 || (reinterpret_cast<uintptr_t>(dst) & 15) != 0)
    return hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, s) == hipSuccess ? RBX_OK : RBX_ERR_LAUNCH;
  const size_t n16 = bytes / 16;
  const int tail = static_cast<int>(bytes - n16 * 16);
  size_t blocks = (n16 + 255) / 256;
  if (blocks > static_cast<size_t>(kCUs) * 16) blocks = static_cast<size_t>(kCUs) * 16;
  if (blocks == 0) blocks = 1;
  hipLaunchKernelGGL(copy_block_kernel, dim3(static_cast<unsigned>(blocks)), dim3(256), 0, s,
                     reinterpret_cast<const uint4*>(src), reinterpret_cast<uint4*>(dst), n16,
                     reinterpret_cast<const unsigned char*>(src + n16 * 16), reinterpret_cast<unsigned char*>(dst + n16 * 16),
                     tail);
  return check_launch("copy_block_kernel");
}
}  // namespace rbx

extern "C" int rbx_comm_bind(void* fn_group_start, void* fn_group_end, void* fn_send, void* fn_recv, void* fn_error_string) {
  using namespace rbx;
  if (!fn_group_start || !fn_group_end || !fn_send || !fn_recv)
    return fail(RBX_ERR_INVALID, "comm_bind: ncclGroupStart / ncclGroupEnd / ncclSend / ncclRecv are required");
  g_group_start = reinterpret_cast<nccl_group_fn>(fn_group_start);
  g_group_end = reinterpret_cast<nccl_group_fn>(fn_group_end);
  g_send = reinterpret_cast<nccl_p2p_fn>(fn_send);
  g_recv = reinterpret_cast<nccl_p2p_fn>(fn_recv);
  g_errstr = reinterpret_cast<nccl_err_fn>(fn_error_string);
  return RBX_OK;
}

extern "C" int rbx_all_to_all(void* comm, const void* d_send, void* d_recv, size_t bytes_per_peer, int32_t world,
                              void* stream) {
  using namespace rbx;
  if (g_send == nullptr) return fail(RBX_ERR_INVALID, "all_to_all: rbx_comm_bind has not been called");
  if (comm == nullptr || world <= 0) return fail(RBX_ERR_INVALID, "all_to_all: no communicator / world %d", world);
  if (bytes_per_peer == 0) return RBX_OK;
  if (d_send == nullptr || d_recv == nullptr) return fail(RBX_ERR_INVALID, "all_to_all: NULL buffer");
  hipStream_t s = as_stream(stream);
  const char* sp = static_cast<const char*>(d_send);
  char* rp = static_cast<char*>(d_recv);
  // The block a rank keeps for itself does not go through RCCL: its self send / recv is a generic copy kernel that moves
  // ~1 TB/s (200 us for the 201 MB block of cfg 3 in a world of one, profiles/r04), a plain copy kernel 3-4x that.  Needs the rank of this process in the communicator (ncclCommUserRank, bound by rbx_comm_bind_collectives).
  int self = -1;
  if (g_userrank != nullptr && g_userrank(comm, &self) != 0) self = -1;
  if (self >= 0 && self < world) {
    if (copy_block(sp + static_cast<size_t>(self) * bytes_per_peer, rp + static_cast<size_t>(self) * bytes_per_peer,
                   bytes_per_peer, s) != RBX_OK)
      return fail(RBX_ERR_LAUNCH, "all_to_all: device-to-device copy of the rank's own block failed");
    if (world == 1) return RBX_OK;
  }
  int rc = g_group_start();
  for (int peer = 0; peer < world && rc == 0; ++peer) {
    if (peer == self) continue;
    rc = g_send(const_cast<char*>(sp) + static_cast<size_t>(peer) * bytes_per_peer, bytes_per_peer, kNcclInt8, peer, comm, s);
    if (rc == 0) rc = g_recv(rp + static_cast<size_t>(peer) * bytes_per_peer, bytes_per_peer, kNcclInt8, peer, comm, s);
  }
  const int rc_end = g_group_end();
  if (rc == 0) rc = rc_end;
  if (rc != 0) return fail(RBX_ERR_LAUNCH, "all_to_all: RCCL error %d (%s)", rc, g_errstr ? g_errstr(rc) : "?");
  return RBX_OK;
}

extern "C" int rbx_comm_bind_collectives(void* fn_all_reduce, void* fn_all_gather, void* fn_comm_user_rank) {
  using namespace rbx;
  if (!fn_all_reduce) return fail(RBX_ERR_INVALID, "comm_bind_collectives: ncclAllReduce is required");
  g_allreduce = reinterpret_cast<nccl_allreduce_fn>(fn_all_reduce);
  g_allgather = reinterpret_cast<nccl_allgather_fn>(fn_all_gather);
  g_userrank = reinterpret_cast<nccl_rank_fn>(fn_comm_user_rank);
  return RBX_OK;
}

extern "C" int rbx_all_reduce(void* comm, const void* d_send, void* d_recv, size_t count, int32_t dtype, int32_t op,
                              void* stream) {
  using namespace rbx;
  if (g_allreduce == nullptr) return fail(RBX_ERR_INVALID, "all_reduce: rbx_comm_bind_collectives has not been called");
  if (comm == nullptr) return fail(RBX_ERR_INVALID, "all_reduce: no communicator");
  const int t = nccl_type(dtype);
  if (t < 0) return fail(RBX_ERR_INVALID, "all_reduce: dtype %d", dtype);
  if (op != RBX_REDUCE_SUM && op != RBX_REDUCE_MAX && op != RBX_REDUCE_MIN)
    return fail(RBX_ERR_INVALID, "all_reduce: op %d", op);
  if (count == 0) return RBX_OK;
  if (d_send == nullptr || d_recv == nullptr) return fail(RBX_ERR_INVALID, "all_reduce: NULL buffer");
  // RBX_REDUCE_SUM / MAX / MIN carry ncclRedOp_t's own values (ncclSum = 0, ncclMax = 2, ncclMin = 3)
  const int rc = g_allreduce(d_send, d_recv, count, t, op, comm, as_stream(stream));
  if (rc != 0) return fail(RBX_ERR_LAUNCH, "all_reduce: RCCL error %d (%s)", rc, g_errstr ? g_errstr(rc) : "?");
  return RBX_OK;
}

extern "C" int rbx_all_gather(void* comm, const void* d_send, void* d_recv, size_t bytes_per_rank, void* stream) {
  using namespace rbx;
  if (g_allgather == nullptr) return fail(RBX_ERR_INVALID, "all_gather: rbx_comm_bind_collectives has not been given ncclAllGather");
  if (comm == nullptr) return fail(RBX_ERR_INVALID, "all_gather: no communicator");
  if (bytes_per_rank == 0) return RBX_OK;
  if (d_send == nullptr || d_recv == nullptr) return fail(RBX_ERR_INVALID, "all_gather: NULL buffer");
  const int rc = g_allgather(d_send, d_recv, bytes_per_rank, kNcclInt8, comm, as_stream(stream));
  if (rc != 0) return fail(RBX_ERR_LAUNCH, "all_gather: RCCL error %d (%s)", rc, g_errstr ? g_errstr(rc) : "?");
  return RBX_OK;
}
