// capi_rccl.hip -- the transports of a multi-GPU run behind the C boundary: librccl loaded on first use, the reduction over the ranks
// (hook or native ncclAllReduce on the context's stream), the probe exchange, and the gather / extract of surfel shards.
#include "capi_internal.h"

using namespace bahip;
using namespace bahip_capi;

namespace bahip_capi {
// ---- RCCL, loaded on first use ---------------------------------------------------------------------------------------
// The prototypes below restate the four RCCL entry points used (rccl.h: ncclGetUniqueId, ncclCommInitRank, ncclAllReduce,
// ncclCommDestroy, ncclGetErrorString); ncclFloat = 7, ncclInt64 = 4, ncclSum = 0 in every NCCL / RCCL release.
struct RcclId { char internal[BAHIP_RCCL_UNIQUE_ID_BYTES]; };
struct RcclApi {
  void* handle = nullptr;
  int (*GetUniqueId)(void*) = nullptr;
  int (*CommInitRank)(void**, int, RcclId /* ncclUniqueId, by value */, int) = nullptr;
  int (*AllReduce)(const void*, void*, size_t, int, int, void*, hipStream_t) = nullptr;
  int (*CommDestroy)(void*) = nullptr;
  const char* (*GetErrorString)(int) = nullptr;
};
RcclApi g_rccl;
int load_rccl() {
  if (g_rccl.handle) return 0;
  void* h = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
  if (!h) h = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
  if (!h) h = dlopen("/opt/rocm/lib/librccl.so", RTLD_NOW | RTLD_GLOBAL);
  if (!h) return fail("librccl.so could not be loaded (multi-GPU needs RCCL)", __FILE__, __LINE__);
  g_rccl.GetUniqueId = reinterpret_cast<decltype(g_rccl.GetUniqueId)>(dlsym(h, "ncclGetUniqueId"));
  g_rccl.CommInitRank = reinterpret_cast<decltype(g_rccl.CommInitRank)>(dlsym(h, "ncclCommInitRank"));
  g_rccl.AllReduce = reinterpret_cast<decltype(g_rccl.AllReduce)>(dlsym(h, "ncclAllReduce"));
  g_rccl.CommDestroy = reinterpret_cast<decltype(g_rccl.CommDestroy)>(dlsym(h, "ncclCommDestroy"));
  g_rccl.GetErrorString = reinterpret_cast<decltype(g_rccl.GetErrorString)>(dlsym(h, "ncclGetErrorString"));
  if (!g_rccl.GetUniqueId || !g_rccl.CommInitRank || !g_rccl.AllReduce || !g_rccl.CommDestroy)
    return fail("librccl.so lacks ncclGetUniqueId / ncclCommInitRank / ncclAllReduce / ncclCommDestroy", __FILE__, __LINE__);
  g_rccl.handle = h;
  return 0;
}
int rccl_fail(const char* what, int rc) {
  char buf[256];
  snprintf(buf, sizeof(buf), "%s failed: %s", what, g_rccl.GetErrorString ? g_rccl.GetErrorString(rc) : "RCCL error");
  g_last_error = buf;
  return 1;
}
int rccl_allreduce(bahip_context* ctx, void* buffer, size_t count, int dtype) {
  const int nccl_type = dtype == BAHIP_SUM_I64 ? 4 /* ncclInt64 */ : dtype == BAHIP_SUM_F64 ? 8 /* ncclDouble */ : 7 /* ncclFloat */;
  const int rc = g_rccl.AllReduce(buffer, buffer, count, nccl_type, 0 /* ncclSum */, ctx->rccl_comm, ctx->stream);
  return rc == 0 ? 0 : rccl_fail("ncclAllReduce", rc);
}

// Element-wise sum of a device buffer over all ranks, in place, ordered on the context's stream: the caller's hook if one is
// installed (it overrides: a caller that installs a hook after bahip_context_init_rccl wants the hook), else the native RCCL
// path if a communicator exists, else nothing (single GPU).
int reduce_over_ranks(bahip_context* ctx, void* buffer, size_t count, int dtype) {
  if (count == 0) return 0;
  if (ctx->allreduce || ctx->rccl_comm) {
    ctx->exchange_calls += 1;
    ctx->exchange_bytes += (long long)count * (dtype == BAHIP_SUM_F32 ? 4 : 8);
  }
  if (ctx->allreduce) {
    if (ctx->allreduce(buffer, count, dtype, ctx->stream, ctx->allreduce_user) != 0) return fail("all-reduce hook failed", __FILE__, __LINE__);
    return 0;
  }
  if (ctx->rccl_comm) return rccl_allreduce(ctx, buffer, count, dtype);
  return 0;
}


void rccl_destroy_communicator(bahip_context* ctx) {
  if (ctx->rccl_comm && g_rccl.CommDestroy) g_rccl.CommDestroy(ctx->rccl_comm);
  ctx->rccl_comm = nullptr;
}
}  // namespace bahip_capi

extern "C" {
int bahip_context_is_sharded(bahip_context* ctx) { return (ctx->allreduce != nullptr || ctx->rccl_comm != nullptr) ? 1 : 0; }

int bahip_context_set_allreduce(bahip_context* ctx, bahip_allreduce_fn fn, void* user) {
  ctx->allreduce = fn;
  ctx->allreduce_user = user;
  return 0;
}

int bahip_context_set_sum_classes(bahip_context* ctx, int classes) {
  REQUIRE(classes == 4 || classes == 8, "the per-surfel sums of the normals / geometry passes are defined over 4 or 8 keyframe classes");
  REQUIRE(ctx->kf_world <= classes, "keyframe sharding over more ranks than classes");
  ctx->sum_classes = classes;
  ctx->in.sum_classes = classes;
  return 0;
}

int bahip_context_set_keyframe_sharding(bahip_context* ctx, int rank, int world) {
  REQUIRE(world == 1 || world == 2 || world == 4 || world == 8, "keyframe sharding: world must be 1, 2, 4 or 8 (a rank holds whole keyframe classes)");
  REQUIRE(world <= ctx->sum_classes, "keyframe sharding over 8 ranks needs the 8-class definition of the per-surfel sums: bahip_context_set_sum_classes(ctx, 8) "
                                     "first (on the single-GPU run it is compared with as well: the class count is part of the sums' definition)");
  REQUIRE(rank >= 0 && rank < world, "keyframe sharding: rank out of range");
  ctx->kf_rank = rank; ctx->kf_world = world;
  return 0;
}

int bahip_rccl_get_unique_id(char unique_id_out[BAHIP_RCCL_UNIQUE_ID_BYTES]) {
  REQUIRE(unique_id_out != nullptr, "bahip_rccl_get_unique_id: NULL argument");
  if (load_rccl()) return 1;
  RcclId id;
  const int rc = g_rccl.GetUniqueId(&id);
  if (rc != 0) return rccl_fail("ncclGetUniqueId", rc);
  memcpy(unique_id_out, id.internal, BAHIP_RCCL_UNIQUE_ID_BYTES);
  return 0;
}

int bahip_context_init_rccl(bahip_context* ctx, const char unique_id[BAHIP_RCCL_UNIQUE_ID_BYTES], int rank, int world_size) {
  REQUIRE(ctx != nullptr && unique_id != nullptr, "bahip_context_init_rccl: NULL argument");
  REQUIRE(world_size >= 1 && rank >= 0 && rank < world_size, "bahip_context_init_rccl: rank / world_size out of range");
  if (load_rccl()) return 1;
  if (ctx->rccl_comm) { g_rccl.CommDestroy(ctx->rccl_comm); ctx->rccl_comm = nullptr; }
  RcclId id;
  memcpy(id.internal, unique_id, BAHIP_RCCL_UNIQUE_ID_BYTES);
  void* comm = nullptr;
  const int rc = g_rccl.CommInitRank(&comm, world_size, id, rank);
  if (rc != 0) return rccl_fail("ncclCommInitRank", rc);
  ctx->rccl_comm = comm;
  ctx->world = world_size;
  return 0;
}

// The first exchange of a run, as a probe: every rank contributes 1 through whatever transport the context uses (the hook or the
// native RCCL communicator), on the context's stream, and the host waits for the sum with a time limit.  A multi-rank job whose
// collective cannot complete (a rank that never arrived, a fabric that does not come up) otherwise hangs in the first BA
// iteration without a word; this returns an error that says which exchange it was and how long it waited.
int bahip_context_count_ranks(bahip_context* ctx, int timeout_ms, int* ranks_out) {
  REQUIRE(ctx != nullptr && ranks_out != nullptr, "bahip_context_count_ranks: NULL argument");
  *ranks_out = 1;
  if (!is_sharded(ctx)) return 0;
  DevMem word;
  HIP_TRY(hipMalloc(&word.p, sizeof(long long)));
  const long long one = 1;
  HIP_TRY(hipMemcpyAsync(word.p, &one, sizeof(one), hipMemcpyHostToDevice, ctx->stream));
  if (reduce_over_ranks(ctx, word.p, 1, BAHIP_SUM_I64)) return 1;
  hipEvent_t done;
  HIP_TRY(hipEventCreateWithFlags(&done, hipEventDisableTiming));
  HIP_TRY(hipEventRecord(done, ctx->stream));
  const auto start = std::chrono::steady_clock::now();
  for (;;) {
    const hipError_t state = hipEventQuery(done);
    if (state == hipSuccess) break;
    if (state != hipErrorNotReady) { hipEventDestroy(done); return fail("the probe exchange failed on the device", __FILE__, __LINE__); }
    if (timeout_ms > 0 && std::chrono::duration_cast<std::chrono::milliseconds>(std::chrono::steady_clock::now() - start).count() > timeout_ms) {
      // (the event and the buffer are left alone: the collective may still own them)
      word.p = nullptr;
      return fail(ctx->allreduce ? "the first all-reduce (hook transport) did not complete within the time limit: not every rank reached it"
                                 : "the first ncclAllReduce (native RCCL transport over xGMI) did not complete within the time limit: not every rank "
                                   "reached it, or the communicator's links did not come up (NCCL_DEBUG=INFO shows the ring)", __FILE__, __LINE__);
    }
    std::this_thread::sleep_for(std::chrono::milliseconds(1));
  }
  HIP_TRY(hipEventDestroy(done));
  long long seen = 0;
  HIP_TRY(hipMemcpy(&seen, word.p, sizeof(seen), hipMemcpyDeviceToHost));
  *ranks_out = (int)seen;
  return 0;
}

int bahip_exchange_stats(bahip_context* ctx, long long* calls_out, long long* bytes_out, int reset) {
  if (calls_out) *calls_out = ctx->exchange_calls;
  if (bytes_out) *bytes_out = ctx->exchange_bytes;
  if (reset) { ctx->exchange_calls = 0; ctx->exchange_bytes = 0; }
  return 0;
}

}  // extern "C"
namespace {
float* host_row(const SurfelsView& v, int row) { return reinterpret_cast<float*>(reinterpret_cast<char*>(v.data) + (size_t)row * v.pitch); }
// number of surfels of a cloud of `total` that the chunk-cyclic partition gives to `rank`
uint32_t shard_size_of(uint32_t total, int rank, int world, uint32_t chunk) {
  const uint64_t stride = (uint64_t)chunk * (uint64_t)world;
  const uint64_t full = total / stride, rest = total % stride;
  const uint64_t begin = (uint64_t)rank * chunk;
  const uint64_t tail = rest > begin ? (rest - begin < chunk ? rest - begin : chunk) : 0;
  return (uint32_t)(full * chunk + tail);
}
}  // namespace
extern "C" {

int bahip_gather_surfel_shards(bahip_context* ctx, const bahip_surfels* shard, uint32_t shard_surfel_count, int rank, int world, uint32_t chunk,
                               bahip_surfels* cloud, uint32_t* cloud_surfels_size_out, uint32_t* cloud_surfel_count_out) {
  REQUIRE(world >= 1 && rank >= 0 && rank < world && chunk > 0 && chunk % 64 == 0, "bahip_gather_surfel_shards: bad partition (chunks are whole 64-surfel tiles)");
  REQUIRE(world <= 64, "bahip_gather_surfel_shards: at most 64 ranks");
  hipStream_t st = ctx->stream;
  // every rank's (size, count): a sum over the ranks of a table that is zero except for the own row
  long long table[128] = {0};
  table[2 * rank] = shard->surfels_size; table[2 * rank + 1] = shard_surfel_count;
  DevMem dev_table;
  HIP_TRY(hipMalloc(&dev_table.p, sizeof(table)));
  HIP_TRY(hipMemcpyAsync(dev_table.p, table, sizeof(table), hipMemcpyHostToDevice, st));
  if (reduce_over_ranks(ctx, dev_table.p, 2 * (size_t)world, BAHIP_SUM_I64)) return 1;
  HIP_TRY(hipMemcpyAsync(table, dev_table.p, sizeof(table), hipMemcpyDeviceToHost, st));
  HIP_TRY(hipStreamSynchronize(st));
  uint64_t total = 0, count = 0;
  for (int r = 0; r < world; ++r) { total += (uint64_t)table[2 * r]; count += (uint64_t)table[2 * r + 1]; }
  REQUIRE(is_sharded(ctx) || world == 1, "bahip_gather_surfel_shards: world > 1 needs a communicator or an all-reduce hook");
  REQUIRE(total <= cloud->capacity, "bahip_gather_surfel_shards: the cloud buffer is too small for the union of the shards");
  REQUIRE(cloud->capacity % 8 == 0, "bahip_gather_surfel_shards: the cloud's capacity must be a multiple of 8 (rows travel as 64-bit words)");
  for (int r = 0; r < world; ++r)
    REQUIRE((uint64_t)table[2 * r] == shard_size_of((uint32_t)total, r, world, chunk),
            "bahip_gather_surfel_shards: the shards are not the chunk-cyclic partition of one cloud");
  cloud->surfels_size = (uint32_t)total;
  const SurfelsView sv = make_view(shard), cv = make_view(cloud);
  const size_t words = ((size_t)total + 1) / 2;   // int64 words per data row (rows start 8-byte aligned: pitched allocations)
  for (int row = 0; row < kSurfelAccum0; ++row) HIP_TRY(hipMemsetAsync(host_row(cv, row), 0, words * 8, st));
  if (cv.active) HIP_TRY(hipMemsetAsync(cv.active, 0, ((size_t)total + 7) / 8 * 8, st));
  launch_shard_to_cloud(st, sv, cv, (uint32_t)rank, (uint32_t)world, chunk);
  CHECK_LAUNCH();
  for (int row = 0; row < kSurfelAccum0; ++row)
    if (reduce_over_ranks(ctx, host_row(cv, row), words, BAHIP_SUM_I64)) return 1;
  if (cv.active && reduce_over_ranks(ctx, cv.active, ((size_t)total + 7) / 8, BAHIP_SUM_I64)) return 1;
  if (cloud_surfels_size_out) *cloud_surfels_size_out = (uint32_t)total;
  if (cloud_surfel_count_out) *cloud_surfel_count_out = (uint32_t)count;
  return 0;
}

int bahip_extract_surfel_shard(bahip_context* ctx, const bahip_surfels* cloud, int rank, int world, uint32_t chunk, bahip_surfels* shard,
                               uint32_t* shard_surfels_size_out) {
  REQUIRE(world >= 1 && rank >= 0 && rank < world && chunk > 0 && chunk % 64 == 0, "bahip_extract_surfel_shard: bad partition");
  const uint32_t mine = shard_size_of(cloud->surfels_size, rank, world, chunk);
  REQUIRE(mine <= shard->capacity, "bahip_extract_surfel_shard: the shard buffer is too small");
  shard->surfels_size = mine;
  launch_cloud_to_shard(ctx->stream, make_view(cloud), make_view(shard), (uint32_t)rank, (uint32_t)world, chunk);
  CHECK_LAUNCH();
  if (shard_surfels_size_out) *shard_surfels_size_out = mine;
  return 0;
}

}  // extern "C"
