// dr_comm.cpp -- the multi-GPU half of the C ABI (include/dartray_hip.h): one process per GPU, the per-rank films
// merged by ONE ncclReduce(sum, f32) over xGMI.
//
// The reference fans a render out over isolates and merges their rectangles in the host
// (lib/dartray_web/render_manager.dart:100-141; GetSubWindow lib/core/common.dart:52-73).  Here the merge is a
// collective on device memory, so a foreign host (Dart over dart:ffi, a C program) shards a render with nothing
// but these entry points and a way to carry 128 bytes from rank 0 to the other ranks.
//
// librccl is bound at run time (dlopen), not at link time: a single-GPU host does not need it, and inside a
// Python process that has imported torch the copy torch already mapped is reused (two RCCL copies in one process
// would each bring their own bootstrap threads; two HIP runtimes would not see the device at all).
#include <dlfcn.h>
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>

#include "../../include/dartray_hip.h"

#include "dr_options.h"  // dr_opt: dr_set_option's value, else the environment's (by value)

int dr_fail(int code, const std::string& msg);  // dr_api.hip: sets dr_last_error()

namespace {

// The subset of rccl.h this file uses (RCCL 2.x ABI: ncclUniqueId is 128 opaque bytes passed BY VALUE).
struct NcclUniqueId {
  char internal[DR_COMM_ID_BYTES];
};
typedef void* NcclComm;
enum { kNcclFloat32 = 7, kNcclFloat64 = 8 };  // ncclDataType_t
enum { kNcclSum = 0, kNcclMax = 2 };          // ncclRedOp_t
}  // namespace

// The hand-declared subset above is checked against the installed header at BUILD time (declarations only: nothing of
// librccl is linked), and the library bound at run time must be the same major version.
#if __has_include(<rccl/rccl.h>)
#include <rccl/rccl.h>
static_assert((int)ncclFloat32 == kNcclFloat32 && (int)ncclFloat64 == kNcclFloat64, "ncclDataType_t values changed");
static_assert((int)ncclSum == kNcclSum && (int)ncclMax == kNcclMax, "ncclRedOp_t values changed");
static_assert(sizeof(ncclUniqueId) == DR_COMM_ID_BYTES && NCCL_UNIQUE_ID_BYTES == DR_COMM_ID_BYTES, "ncclUniqueId is not DR_COMM_ID_BYTES long");
static_assert(sizeof(ncclComm_t) == sizeof(void*), "ncclComm_t is not a pointer");
#define DR_RCCL_BUILD_MAJOR NCCL_MAJOR
#else
#define DR_RCCL_BUILD_MAJOR 2
#endif

namespace {

struct Rccl {
  void* handle = nullptr;
  int (*GetUniqueId)(NcclUniqueId*) = nullptr;
  int (*CommInitRank)(NcclComm*, int, NcclUniqueId, int) = nullptr;
  int (*CommDestroy)(NcclComm) = nullptr;
  int (*Reduce)(const void*, void*, size_t, int, int, int, NcclComm, hipStream_t) = nullptr;
  int (*AllReduce)(const void*, void*, size_t, int, int, NcclComm, hipStream_t) = nullptr;
  const char* (*GetErrorString)(int) = nullptr;
  int (*GetVersion)(int*) = nullptr;
};

Rccl g_rccl;
NcclComm g_comm = nullptr;
int g_rank = -1, g_world = 0;

int loadRccl() {
  if (g_rccl.handle) return DR_OK;
  void* h = nullptr;
  std::string tried;
  const DrOpt env = dr_opt("DARTRAY_RCCL_LIB");
  if (env && !env.value.empty()) {
    h = dlopen(env.value.c_str(), RTLD_NOW | RTLD_GLOBAL);
    if (!h) return dr_fail(DR_ERR_UNSUPPORTED, std::string("DARTRAY_RCCL_LIB: ") + dlerror());
  }
  // a copy that is already mapped (torch's) first, then the system one
  const char* names[] = {"librccl.so.1", "librccl.so"};
  for (int pass = 0; pass < 2 && !h; ++pass)
    for (const char* n : names) {
      h = dlopen(n, RTLD_NOW | RTLD_GLOBAL | (pass == 0 ? RTLD_NOLOAD : 0));
      if (h) break;
      if (pass == 1) tried += std::string(" ") + n + ": " + dlerror() + ";";
    }
  if (!h) {
    h = dlopen("/opt/rocm/lib/librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
    if (!h) return dr_fail(DR_ERR_UNSUPPORTED, "librccl not found (set DARTRAY_RCCL_LIB):" + tried);
  }
  Rccl r;
  r.handle = h;
#define BIND(field, sym)                                                                           \
  do {                                                                                             \
    *(void**)(&r.field) = dlsym(h, sym);                                                           \
    if (!r.field) return dr_fail(DR_ERR_UNSUPPORTED, std::string("librccl lacks ") + sym);        \
  } while (0)
  BIND(GetUniqueId, "ncclGetUniqueId");
  BIND(CommInitRank, "ncclCommInitRank");
  BIND(CommDestroy, "ncclCommDestroy");
  BIND(Reduce, "ncclReduce");
  BIND(AllReduce, "ncclAllReduce");
  BIND(GetErrorString, "ncclGetErrorString");
  BIND(GetVersion, "ncclGetVersion");
#undef BIND
  {  // NCCL_VERSION_CODE = major * 10000 + minor * 100 + patch since 2.9 (major * 1000 + ... before): same major as the header
    int v = 0;
    if (r.GetVersion(&v) != 0) return dr_fail(DR_ERR_UNSUPPORTED, "ncclGetVersion failed");
    const int major = v >= 10000 ? v / 10000 : v / 1000;
    if (major != DR_RCCL_BUILD_MAJOR)
      return dr_fail(DR_ERR_UNSUPPORTED, "librccl major version " + std::to_string(major) + " differs from the ABI subset this library declares (" +
                                             std::to_string(DR_RCCL_BUILD_MAJOR) + ")");
  }
  g_rccl = r;
  return DR_OK;
}

int ncclFail(const char* what, int rc) {
  return dr_fail(DR_ERR_HIP, std::string(what) + ": " + (g_rccl.GetErrorString ? g_rccl.GetErrorString(rc) : "?"));
}

}  // namespace

extern "C" {

int dr_comm_available(void) { return loadRccl(); }

int dr_comm_unique_id(void* id_out, uint64_t cap) {
  if (!id_out || cap < DR_COMM_ID_BYTES) return dr_fail(DR_ERR_INVALID, "id buffer smaller than DR_COMM_ID_BYTES");
  int rc = loadRccl();
  if (rc) return rc;
  NcclUniqueId id;
  memset(&id, 0, sizeof(id));
  int n = g_rccl.GetUniqueId(&id);
  if (n) return ncclFail("ncclGetUniqueId", n);
  memcpy(id_out, &id, sizeof(id));
  return DR_OK;
}

int dr_comm_init(int32_t rank, int32_t world, const void* unique_id, uint64_t id_bytes) {
  if (g_comm) return dr_fail(DR_ERR_INVALID, "dr_comm_init: a communicator already exists (dr_comm_destroy first)");
  if (world < 1 || rank < 0 || rank >= world) return dr_fail(DR_ERR_INVALID, "dr_comm_init: rank / world out of range");
  if (!unique_id || id_bytes != DR_COMM_ID_BYTES) return dr_fail(DR_ERR_INVALID, "dr_comm_init: unique id must be DR_COMM_ID_BYTES long");
  int dev = -1;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0) return dr_fail(DR_ERR_NO_DEVICE, "dr_comm_init before dr_init");
  int rc = loadRccl();
  if (rc) return rc;
  NcclUniqueId id;
  memcpy(&id, unique_id, sizeof(id));
  NcclComm c = nullptr;
  int n = g_rccl.CommInitRank(&c, world, id, rank);
  if (n) return ncclFail("ncclCommInitRank", n);
  g_comm = c;
  g_rank = rank;
  g_world = world;
  return DR_OK;
}

int dr_film_reduce(void* film_dev, int64_t npixels, int32_t root, void* hip_stream) {
  if (!g_comm) return dr_fail(DR_ERR_INVALID, "dr_film_reduce before dr_comm_init");
  if (!film_dev || npixels < 0 || root < 0 || root >= g_world) return dr_fail(DR_ERR_INVALID, "dr_film_reduce: bad argument");
  if (npixels == 0) return DR_OK;
  // in place on the root (sendbuff == recvbuff); recvbuff is ignored on the other ranks
  int n = g_rccl.Reduce(film_dev, film_dev, (size_t)npixels * 4, kNcclFloat32, kNcclSum, root, g_comm, (hipStream_t)hip_stream);
  if (n) return ncclFail("ncclReduce", n);
  return DR_OK;
}

int dr_comm_allreduce_f64(void* buf_dev, int64_t n, int32_t op_max, void* hip_stream) {
  if (!g_comm) return dr_fail(DR_ERR_INVALID, "dr_comm_allreduce_f64 before dr_comm_init");
  if (!buf_dev || n <= 0) return dr_fail(DR_ERR_INVALID, "dr_comm_allreduce_f64: bad argument");
  int rc = g_rccl.AllReduce(buf_dev, buf_dev, (size_t)n, kNcclFloat64, op_max ? kNcclMax : kNcclSum, g_comm, (hipStream_t)hip_stream);
  if (rc) return ncclFail("ncclAllReduce", rc);
  return DR_OK;
}

int dr_comm_rank(void) { return g_rank; }
int dr_comm_world(void) { return g_world; }

int dr_comm_destroy(void) {
  if (!g_comm) return DR_OK;
  (void)hipDeviceSynchronize();
  int n = g_rccl.CommDestroy(g_comm);
  g_comm = nullptr;
  g_rank = -1;
  g_world = 0;
  if (n) return ncclFail("ncclCommDestroy", n);
  return DR_OK;
}

}  // extern "C"
