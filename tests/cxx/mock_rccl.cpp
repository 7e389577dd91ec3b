// A test double for RCCL (tests only; never shipped, never loaded by a product path unless the
// test hook AVR_RCCL_LIBRARY names it): the eight entry points csrc/avr_comm.cpp binds with dlsym,
// with the MATCHING semantics of grouped point-to-point operations -- so that the branch of the
// communicator that talks to N > 1 RCCL ranks (the frame's grouped round with its gather rider,
// the classic direct send, the in-band control plane) executes for real where RCCL itself cannot
// place two ranks on one device: the ranks are host threads of ONE process on one GPU.
//
// Semantics kept: operations between a pair of ranks match in program order (the k-th send of a to
// b with the k-th receive of b from a); a group's operations complete together; counts must agree
// (RCCL would hang on a mismatch -- the double reports it: ncclInvalidUsage, and says which);
// every rank of the communicator has to reach its ncclGroupEnd (a missing one is a hang in RCCL:
// here the wait gives up after 20 s with ncclSystemError).  Not kept: asynchrony (a group end
// drains the ranks' streams and copies on the host's command) and speed.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <chrono>
#include <condition_variable>
#include <cstdio>
#include <cstring>
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <vector>

namespace {

struct Op {
  bool send;
  const void* src;
  void* dst;
  size_t bytes;
  int peer;
};

struct World {
  int n = 0;
  std::mutex mutex;
  std::condition_variable arrived;
  int waiting = 0, joined = 0;
  unsigned long generation = 0;
  bool broken = false;
  std::vector<std::vector<Op>> ops;  // per rank: the group being ended
  std::string failure;

  bool barrier() {  // false: a peer did not arrive
    std::unique_lock<std::mutex> lock(mutex);
    if (broken) return false;
    const unsigned long mine = generation;
    if (++waiting == n) {
      waiting = 0;
      ++generation;
      arrived.notify_all();
      return true;
    }
    if (!arrived.wait_for(lock, std::chrono::seconds(20), [&] { return generation != mine || broken; })) {
      broken = true;
      arrived.notify_all();
    }
    return generation != mine;
  }
};

std::mutex g_worlds_mutex;
std::map<std::string, std::shared_ptr<World>> g_worlds;

size_t size_of(ncclDataType_t type) {
  switch (type) {
    case ncclInt8:
    case ncclUint8:
      return 1;
    case ncclFloat16:
    case ncclBfloat16:
      return 2;
    case ncclInt32:
    case ncclUint32:
    case ncclFloat32:
      return 4;
    default:
      return 8;
  }
}

thread_local int t_depth = 0;
thread_local std::vector<Op> t_ops;
thread_local std::vector<hipStream_t> t_streams;
thread_local struct ncclComm* t_comm = nullptr;

}  // namespace

struct ncclComm {
  std::shared_ptr<World> world;
  int rank = 0;
};

extern "C" {

ncclResult_t ncclGetUniqueId(ncclUniqueId* id) {
  static std::mutex mutex;
  static unsigned long counter = 0;
  std::lock_guard<std::mutex> lock(mutex);
  std::memset(id->internal, 0, sizeof(id->internal));
  std::snprintf(id->internal, sizeof(id->internal), "mock-rccl-%lu-%p", ++counter, static_cast<void*>(&counter));
  return ncclSuccess;
}

ncclResult_t ncclCommInitRank(ncclComm_t* comm, int nranks, ncclUniqueId id, int rank) {
  if (comm == nullptr || nranks < 1 || rank < 0 || rank >= nranks) return ncclInvalidArgument;
  std::shared_ptr<World> world;
  {
    std::lock_guard<std::mutex> lock(g_worlds_mutex);
    const std::string key(id.internal, sizeof(id.internal));
    auto& slot = g_worlds[key];
    if (!slot) {
      slot = std::make_shared<World>();
      slot->n = nranks;
      slot->ops.resize(static_cast<size_t>(nranks));
    }
    world = slot;
    if (world->n != nranks) return ncclInvalidArgument;
  }
  auto* out = new ncclComm();
  out->world = world;
  out->rank = rank;
  if (!world->barrier()) {  // collective, as the real one
    delete out;
    return ncclSystemError;
  }
  *comm = out;
  return ncclSuccess;
}

ncclResult_t ncclCommDestroy(ncclComm_t comm) {
  delete comm;
  return ncclSuccess;
}

const char* ncclGetErrorString(ncclResult_t result) {
  switch (result) {
    case ncclSuccess:
      return "no error";
    case ncclInvalidUsage:
      return "invalid usage (mock RCCL: the two sides of an operation disagree -- see stderr)";
    case ncclSystemError:
      return "system error (mock RCCL: a rank did not reach its group end)";
    case ncclInvalidArgument:
      return "invalid argument";
    default:
      return "error";
  }
}

ncclResult_t ncclGroupStart() {
  ++t_depth;
  return ncclSuccess;
}

static ncclResult_t queue_op(const Op& op, ncclComm_t comm, hipStream_t stream) {
  if (comm == nullptr || op.peer < 0 || op.peer >= comm->world->n) return ncclInvalidArgument;
  if (t_comm != nullptr && t_comm != comm) return ncclInvalidUsage;  // one communicator per group here
  t_comm = comm;
  t_ops.push_back(op);
  bool known = false;
  for (hipStream_t s : t_streams) known = known || s == stream;
  if (!known) t_streams.push_back(stream);
  if (t_depth == 0) {  // an operation outside a group is a group of one
    ++t_depth;
    return ncclGroupEnd();
  }
  return ncclSuccess;
}

ncclResult_t ncclSend(const void* buffer, size_t count, ncclDataType_t type, int peer, ncclComm_t comm,
                      hipStream_t stream) {
  return queue_op(Op{true, buffer, nullptr, count * size_of(type), peer}, comm, stream);
}

ncclResult_t ncclRecv(void* buffer, size_t count, ncclDataType_t type, int peer, ncclComm_t comm,
                      hipStream_t stream) {
  return queue_op(Op{false, nullptr, buffer, count * size_of(type), peer}, comm, stream);
}

ncclResult_t ncclGroupEnd() {
  if (t_depth <= 0) return ncclInvalidUsage;
  if (--t_depth > 0) return ncclSuccess;
  ncclComm_t comm = t_comm;
  std::vector<Op> mine;
  mine.swap(t_ops);
  std::vector<hipStream_t> streams;
  streams.swap(t_streams);
  t_comm = nullptr;
  if (comm == nullptr) return ncclSuccess;  // an empty group
  World& world = *comm->world;
  const int me = comm->rank;
  // what the operations read has been produced on their streams
  for (hipStream_t s : streams) {
    if (hipStreamSynchronize(s) != hipSuccess) return ncclUnhandledCudaError;
  }
  {
    std::lock_guard<std::mutex> lock(world.mutex);
    world.ops[static_cast<size_t>(me)] = mine;
  }
  if (!world.barrier()) return ncclSystemError;
  // my k-th receive from p <- p's k-th send to me (program order per pair)
  ncclResult_t status = ncclSuccess;
  std::vector<size_t> taken(static_cast<size_t>(world.n), 0);
  for (const Op& op : mine) {
    if (op.send) continue;
    const std::vector<Op>& theirs = world.ops[static_cast<size_t>(op.peer)];
    size_t& cursor = taken[static_cast<size_t>(op.peer)];
    const Op* match = nullptr;
    while (cursor < theirs.size()) {
      const Op& candidate = theirs[cursor++];
      if (candidate.send && candidate.peer == me) {
        match = &candidate;
        break;
      }
    }
    if (match == nullptr) {
      std::fprintf(stderr, "mock RCCL: rank %d receives %zu bytes from rank %d, which sends nothing more\n", me,
                   op.bytes, op.peer);
      status = ncclInvalidUsage;
    } else if (match->bytes != op.bytes) {
      std::fprintf(stderr, "mock RCCL: rank %d receives %zu bytes from rank %d, which sends %zu\n", me, op.bytes,
                   op.peer, match->bytes);
      status = ncclInvalidUsage;
    } else if (op.bytes > 0 &&
               hipMemcpy(op.dst, match->src, op.bytes, hipMemcpyDeviceToDevice) != hipSuccess) {
      status = ncclUnhandledCudaError;
    }
  }
  // (a device-to-device hipMemcpy may return before the copy has run, and the ranks' streams are
  // non-blocking: they do not order themselves after the null stream it runs on -- the kernels that
  // read what was received would race with it)
  if (hipStreamSynchronize(nullptr) != hipSuccess) status = ncclUnhandledCudaError;
  // every send of mine must have been wanted: count the peers' receives from me
  for (int p = 0; p < world.n; ++p) {
    size_t sends = 0, wanted = 0;
    for (const Op& op : mine) sends += (op.send && op.peer == p) ? 1 : 0;
    for (const Op& op : world.ops[static_cast<size_t>(p)]) wanted += (!op.send && op.peer == me) ? 1 : 0;
    if (sends != wanted) {
      std::fprintf(stderr, "mock RCCL: rank %d sends %zu messages to rank %d, which receives %zu\n", me, sends, p,
                   wanted);
      status = ncclInvalidUsage;
    }
  }
  if (!world.barrier()) return ncclSystemError;  // everybody has copied: the buffers may change
  return status;
}

}  // extern "C"
