// Rank communicator of the sort-last compositor: the image-fragment exchange and the gather of
// DirectSend as collectives on the GPUs' own interconnect (include/avr_hip.h, "rank
// communicator").
//
// Reference: DirectSend/Base/DirectSendBase.cpp:76-177 (PostReceives / PostSends: per run,
// N(N-1) MPI_Isend / MPI_Irecv pairs of host buffers) and Common/ImageColorOnly.hpp:220-270
// (Gather: two MPI_Gather of ints + MPI_Gatherv of bytes).  Here: ONE grouped ncclSend / ncclRecv
// round per frame over RCCL (xGMI, device buffers, on the compositing stream), block sizes known
// to every rank from the replicated frame plan, so no metadata messages exist.
//
// RCCL is bound at run time (dlopen): a single-rank user never loads it, and inside a Python
// process the copy torch already loaded is the one used (one runtime per process).
//
// The "local" flavour connects N rank objects living in ONE process on ONE GPU (one host thread
// per rank): host-synchronous copies between the ranks' buffers.  It exists to rehearse the
// N-rank frame where only one GPU is available (RCCL cannot place two ranks on one device); it is
// not a performance path.
#include <dlfcn.h>
#include <fcntl.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstring>
#include <exception>
#include <memory>
#include <mutex>
#include <stdexcept>
#include <string>
#include <thread>
#include <vector>

#include "avr_internal.h"
#include "avr_plan.h"

namespace avr {

namespace {

void hip_ok(hipError_t err, const char* what) {
  if (err != hipSuccess) throw std::runtime_error(std::string(what) + ": " + hipGetErrorString(err));
}

// ---- RCCL entry points, resolved once ----------------------------------------------------------
struct Rccl {
  void* handle = nullptr;
  decltype(&ncclGetUniqueId) get_unique_id = nullptr;
  decltype(&ncclCommInitRank) comm_init_rank = nullptr;
  decltype(&ncclCommDestroy) comm_destroy = nullptr;
  decltype(&ncclGroupStart) group_start = nullptr;
  decltype(&ncclGroupEnd) group_end = nullptr;
  decltype(&ncclSend) send = nullptr;
  decltype(&ncclRecv) recv = nullptr;
  decltype(&ncclGetErrorString) error_string = nullptr;
};

const Rccl& rccl() {
  static Rccl api;
  static std::once_flag once;
  static std::string failure;
  std::call_once(once, [] {
    // (AVR_RCCL_LIBRARY: a test hook -- tests/cxx/mock_rccl.cpp, a double with RCCL's matching
    // semantics, lets the N > 1 branch below run as rank threads on one GPU)
    if (const char* forced = std::getenv("AVR_RCCL_LIBRARY")) {
      api.handle = dlopen(forced, RTLD_NOW | RTLD_LOCAL);
    } else {
      for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
        api.handle = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
        if (api.handle != nullptr) break;
      }
    }
    if (api.handle == nullptr) {
      failure = std::string("cannot load RCCL (librccl.so.1): ") + dlerror();
      return;
    }
    auto bind = [](void* handle, const char* symbol) {
      void* fn = dlsym(handle, symbol);
      if (fn == nullptr) failure = std::string("RCCL lacks ") + symbol;
      return fn;
    };
    api.get_unique_id = reinterpret_cast<decltype(api.get_unique_id)>(bind(api.handle, "ncclGetUniqueId"));
    api.comm_init_rank = reinterpret_cast<decltype(api.comm_init_rank)>(bind(api.handle, "ncclCommInitRank"));
    api.comm_destroy = reinterpret_cast<decltype(api.comm_destroy)>(bind(api.handle, "ncclCommDestroy"));
    api.group_start = reinterpret_cast<decltype(api.group_start)>(bind(api.handle, "ncclGroupStart"));
    api.group_end = reinterpret_cast<decltype(api.group_end)>(bind(api.handle, "ncclGroupEnd"));
    api.send = reinterpret_cast<decltype(api.send)>(bind(api.handle, "ncclSend"));
    api.recv = reinterpret_cast<decltype(api.recv)>(bind(api.handle, "ncclRecv"));
    api.error_string = reinterpret_cast<decltype(api.error_string)>(bind(api.handle, "ncclGetErrorString"));
  });
  if (!failure.empty()) throw std::runtime_error(failure);
  return api;
}

void nccl_ok(ncclResult_t result, const char* what) {
  if (result != ncclSuccess) {
    throw std::runtime_error(std::string(what) + ": " + rccl().error_string(result));
  }
}

// An open RCCL group is closed on every path out of its scope: an exception between
// ncclGroupStart and ncclGroupEnd (a failed requirement of the gather riding in the round, a HIP
// error) would otherwise leave the group open, and every later RCCL call of the process would
// nest in it and never be launched.
class RcclGroup {
 public:
  explicit RcclGroup(bool open = true) : open_(open) {
    if (open_) nccl_ok(rccl().group_start(), "ncclGroupStart");
  }
  RcclGroup(const RcclGroup&) = delete;
  RcclGroup& operator=(const RcclGroup&) = delete;
  void end() {  // the normal path: errors of the launch itself are reported
    if (!open_) return;
    open_ = false;
    nccl_ok(rccl().group_end(), "ncclGroupEnd");
  }
  ~RcclGroup() {
    if (!open_) return;
    try {
      (void)rccl().group_end();  // unwinding: the first error is the one reported
    } catch (...) {
    }
  }

 private:
  bool open_;
};

// ---- in-process rehearsal communicator -------------------------------------------------------
struct LocalWorld {
  int n_ranks = 0;
  std::mutex mutex;
  std::condition_variable arrived;
  int waiting = 0;
  uint64_t generation = 0;
  // what each rank publishes for the collective in flight
  std::vector<const char*> base;              // send buffer / piece
  std::vector<std::vector<int64_t>> offsets;  // byte offset of the block for each peer
  std::vector<std::vector<int64_t>> sizes;    // byte size of the block for each peer
  std::vector<std::vector<unsigned char>> control;  // avr_comm_control_allgather
  std::vector<std::pair<uint32_t, uint64_t>> tags;  // which collective each rank is in (meet)

  // Every rank arrives, or the wait gives up (AVR_FRAME_TIMEOUT_MS): a rank thread that died must
  // not leave its peers waiting.
  // (a meeting that timed out for one rank is over for all: its count is off for good, so the
  // world is marked broken and every later or pending meeting ends in the same error at once)
  bool broken = false;
  void barrier() {
    std::unique_lock<std::mutex> lock(mutex);
    const int limit_ms = frame_timeout_ms();
    auto gone = [&] {
      return DeadlineExceeded("local communicator: a peer did not arrive within " +
                              std::to_string(limit_ms) + " ms (AVR_FRAME_TIMEOUT_MS)");
    };
    if (broken) throw gone();
    const uint64_t mine = generation;
    if (++waiting == n_ranks) {
      waiting = 0;
      ++generation;
      arrived.notify_all();
      return;
    }
    auto released = [&] { return generation != mine || broken; };
    if (limit_ms <= 0) {
      arrived.wait(lock, released);
    } else if (!arrived.wait_for(lock, std::chrono::milliseconds(limit_ms), released)) {
      broken = true;
      arrived.notify_all();
    }
    if (generation == mine) throw gone();
  }
  // First meeting of a collective: every rank says which one it is in, and all must say the same
  // -- a rehearsal in which an exchange and a control round paired up silently would rehearse
  // nothing (over RCCL such ranks never get out of their rounds: the deadline ends those).
  void meet(int me, uint32_t kind, uint64_t detail);
};

const char* collective_name(uint32_t kind) {
  switch (kind) {
    case 1: return "the frame's exchange";
    case 2: return "the classic direct send";
    case 3: return "the gather";
    case 4: return "a control-plane allgather";
    default: return "an unknown collective";
  }
}

// (every rank reaches this verdict from the same tags: nobody goes on to the collective's second
// meeting, so whoever catches it must not either)
struct CallsDiffer : std::runtime_error {
  using std::runtime_error::runtime_error;
};
[[noreturn]] void calls_differ(int me, uint32_t mine, int other, uint32_t theirs) {
  throw CallsDiffer("the ranks' calls differ: rank " + std::to_string(other) + " is in " +
                           collective_name(theirs) + " while rank " + std::to_string(me) + " is in " +
                           collective_name(mine) + " (the ranks were not driven alike)");
}

void LocalWorld::meet(int me, uint32_t kind, uint64_t detail) {
  {
    std::lock_guard<std::mutex> lock(mutex);
    tags[static_cast<size_t>(me)] = {kind, detail};
  }
  barrier();
  for (int s = 0; s < n_ranks; ++s) {
    for (int t = 0; t < n_ranks; ++t) {  // every rank scans all pairs: the same verdict everywhere
      if (tags[static_cast<size_t>(s)] != tags[static_cast<size_t>(t)]) {
        const int other = (tags[static_cast<size_t>(s)] != tags[static_cast<size_t>(me)]) ? s : t;
        calls_differ(me, kind, other, tags[static_cast<size_t>(other)].first);
      }
    }
  }
}

// ---- cross-process rehearsal communicator --------------------------------------------------
// N rank PROCESSES sharing one GPU meet in a POSIX shared-memory segment: a header with a
// sense-reversing barrier and, per rank, the block offsets / sizes of the collective in flight,
// then one data region per rank.  Blocks travel device -> the sender's region -> device.
struct SharedHeader {
  static constexpr int kMaxRanks = 64;
  static constexpr uint32_t kMagic = 0x41565253u;  // "AVRS": rank 0 has initialised the segment
  std::atomic<uint32_t> magic;
  std::atomic<uint32_t> arrived;
  std::atomic<uint32_t> generation;
  int64_t offsets[kMaxRanks][kMaxRanks];  // [source][peer]: byte offset inside the source's region
  int64_t sizes[kMaxRanks][kMaxRanks];
  unsigned char control[kMaxRanks][AVR_CONTROL_MAX_BYTES];  // avr_comm_control_allgather
  uint64_t tag_kind[kMaxRanks], tag_detail[kMaxRanks];       // which collective each rank is in
  std::atomic<uint32_t> broken;  // a meeting timed out for some rank: over for all, for good
};

struct SharedWorld {
  int n_ranks = 0;
  std::string name;
  void* mapping = nullptr;
  size_t mapped_bytes = 0;
  size_t capacity = 0;  // bytes per rank
  bool owner = false;   // rank 0 removes the name when it goes
  SharedHeader* header() const { return static_cast<SharedHeader*>(mapping); }
  char* region(int rank) const {
    const size_t head = (sizeof(SharedHeader) + 4095) / 4096 * 4096;
    return static_cast<char*>(mapping) + head + static_cast<size_t>(rank) * capacity;
  }
  ~SharedWorld() {
    if (mapping != nullptr) (void)munmap(mapping, mapped_bytes);
    if (owner) (void)shm_unlink(name.c_str());
  }
  // First meeting of a collective (LocalWorld::meet): all ranks must be in the same one.
  void meet(int me, uint32_t kind, uint64_t detail) {
    SharedHeader* h = header();
    h->tag_kind[me] = kind;
    h->tag_detail[me] = detail;
    barrier();
    for (int s = 0; s < n_ranks; ++s) {
      for (int t = 0; t < n_ranks; ++t) {
        if (h->tag_kind[s] != h->tag_kind[t] || h->tag_detail[s] != h->tag_detail[t]) {
          const int other = (h->tag_kind[s] != kind || h->tag_detail[s] != detail) ? s : t;
          calls_differ(me, kind, other, static_cast<uint32_t>(h->tag_kind[other]));
        }
      }
    }
  }
  // Every rank arrives, or the wait gives up: a rehearsal must not hang a GPU box.
  void barrier(int limit_ms = -1) {
    if (limit_ms < 0) limit_ms = frame_timeout_ms();
    SharedHeader* h = header();
    auto gone = [&] {
      return DeadlineExceeded("shared communicator: a peer did not arrive within " +
                              std::to_string(limit_ms) + " ms (AVR_FRAME_TIMEOUT_MS)");
    };
    if (h->broken.load(std::memory_order_acquire) != 0) throw gone();
    const uint32_t mine = h->generation.load(std::memory_order_acquire);
    if (h->arrived.fetch_add(1, std::memory_order_acq_rel) + 1 == static_cast<uint32_t>(n_ranks)) {
      h->arrived.store(0, std::memory_order_relaxed);
      h->generation.fetch_add(1, std::memory_order_release);
      return;
    }
    const auto deadline = std::chrono::steady_clock::now() + std::chrono::milliseconds(limit_ms);
    for (unsigned spins = 0; h->generation.load(std::memory_order_acquire) == mine; ++spins) {
      if (spins > 256) std::this_thread::sleep_for(std::chrono::microseconds(20));
      if ((spins & 255u) == 255u) {
        if (h->broken.load(std::memory_order_acquire) != 0) throw gone();
        if (limit_ms > 0 && std::chrono::steady_clock::now() > deadline) {
          h->broken.store(1, std::memory_order_release);
          throw gone();
        }
      }
    }
  }
};

}  // namespace
}  // namespace avr

struct avr_comm {
  int rank = 0;
  int n_ranks = 1;
  int device = 0;
  ncclComm_t nccl = nullptr;                      // RCCL flavour
  std::shared_ptr<avr::LocalWorld> local;         // in-process flavour
  std::unique_ptr<avr::SharedWorld> shared;       // cross-process flavour (one GPU, shared memory)
  bool solo = false;                              // one rank of N played alone (timing studies)
  int solo_percent = 100;                         // ... share of every peer's block it moves through RCCL
  // control plane (avr_comm_control_allgather): the caller's own (MPI_Allgather in the reference's
  // host, gloo in bench.py), or -- RCCL flavour without one -- a tiny grouped round in band
  avr_control_allgather_fn control_fn = nullptr;
  void* control_user = nullptr;
  void* control_dev = nullptr;     // (n_ranks + 1) slots of AVR_CONTROL_MAX_BYTES: mine, then everybody's
  void* control_host = nullptr;    // pinned twin, device-mapped
  void* control_host_mapped = nullptr;
  hipEvent_t control_done = nullptr;
  long control_rounds = 0;         // calls so far (diagnostics, tests)
};

namespace {

template <typename F>
int guarded(F&& body) {
  const avr::CollectiveScope collective(true);  // a communicator's waits have the default deadline
  try {
    return body();
  } catch (const std::invalid_argument& e) {
    avr::set_error(e.what());
    return AVR_ERR_INVALID_ARGUMENT;
  } catch (const std::exception& e) {
    avr::set_error(e.what());
    return AVR_ERR_RUNTIME;
  } catch (...) {
    avr::set_error("unknown failure");
    return AVR_ERR_RUNTIME;
  }
}

void require(bool condition, const char* message) {
  if (!condition) throw std::invalid_argument(message);
}

void check_plan(const avr_comm* comm, const avr_frame_plan* plan) {
  require(comm != nullptr && plan != nullptr, "null argument");
  require(plan->info.n_ranks == comm->n_ranks && plan->info.rank == comm->rank,
          "the frame plan was made for another rank / rank count than the communicator");
}

// Waits on the host until everything queued on `stream` so far has finished.
void drain(hipStream_t stream) { avr::hip_ok(hipStreamSynchronize(stream), "hipStreamSynchronize"); }

}  // namespace

extern "C" {

int avr_comm_unique_id(char id_out[AVR_COMM_ID_BYTES]) {
  return guarded([&]() -> int {
    require(id_out != nullptr, "null argument");
    static_assert(AVR_COMM_ID_BYTES == NCCL_UNIQUE_ID_BYTES, "id size");
    ncclUniqueId id;
    avr::nccl_ok(avr::rccl().get_unique_id(&id), "ncclGetUniqueId");
    std::memcpy(id_out, id.internal, AVR_COMM_ID_BYTES);
    return AVR_OK;
  });
}

int avr_comm_create(int device_id, const char id[AVR_COMM_ID_BYTES], int rank, int n_ranks,
                    avr_comm** out_comm) {
  return guarded([&]() -> int {
    require(out_comm != nullptr && id != nullptr, "null argument");
    *out_comm = nullptr;
    require(n_ranks >= 1 && rank >= 0 && rank < n_ranks, "invalid rank");
    avr::hip_ok(hipSetDevice(device_id), "hipSetDevice");
    ncclUniqueId unique;
    std::memcpy(unique.internal, id, AVR_COMM_ID_BYTES);
    auto comm = std::make_unique<avr_comm>();
    comm->rank = rank;
    comm->n_ranks = n_ranks;
    comm->device = device_id;
    avr::nccl_ok(avr::rccl().comm_init_rank(&comm->nccl, n_ranks, unique, rank), "ncclCommInitRank");
    *out_comm = comm.release();
    return AVR_OK;
  });
}

int avr_comm_create_local(int n_ranks, avr_comm** out_comms) {
  return guarded([&]() -> int {
    require(out_comms != nullptr && n_ranks >= 1, "invalid argument");
    auto world = std::make_shared<avr::LocalWorld>();
    world->n_ranks = n_ranks;
    world->base.assign(static_cast<size_t>(n_ranks), nullptr);
    world->offsets.resize(static_cast<size_t>(n_ranks));
    world->sizes.resize(static_cast<size_t>(n_ranks));
    world->control.resize(static_cast<size_t>(n_ranks));
    world->tags.assign(static_cast<size_t>(n_ranks), {0u, 0ull});
    for (int r = 0; r < n_ranks; ++r) {
      auto* comm = new avr_comm();
      comm->rank = r;
      comm->n_ranks = n_ranks;
      comm->local = world;
      out_comms[r] = comm;
    }
    return AVR_OK;
  });
}

int avr_comm_create_solo(int rank, int n_ranks, avr_comm** out_comm) {
  return guarded([&]() -> int {
    require(out_comm != nullptr && n_ranks >= 1 && rank >= 0 && rank < n_ranks, "invalid argument");
    auto* comm = new avr_comm();
    comm->rank = rank;
    comm->n_ranks = n_ranks;
    comm->solo = true;
    *out_comm = comm;
    return AVR_OK;
  });
}

int avr_comm_create_solo_rccl(int device_id, int rank, int n_ranks, int percent, avr_comm** out_comm) {
  return guarded([&]() -> int {
    require(out_comm != nullptr && n_ranks >= 1 && rank >= 0 && rank < n_ranks && percent >= 0 &&
                percent <= 100, "invalid argument");
    *out_comm = nullptr;
    avr::hip_ok(hipSetDevice(device_id), "hipSetDevice");
    ncclUniqueId unique;
    avr::nccl_ok(avr::rccl().get_unique_id(&unique), "ncclGetUniqueId");
    auto comm = std::make_unique<avr_comm>();
    comm->rank = rank;
    comm->n_ranks = n_ranks;
    comm->device = device_id;
    comm->solo = true;
    comm->solo_percent = percent;
    avr::nccl_ok(avr::rccl().comm_init_rank(&comm->nccl, 1, unique, 0), "ncclCommInitRank");
    *out_comm = comm.release();
    return AVR_OK;
  });
}

int avr_comm_create_shared(const char* name, int rank, int n_ranks, size_t capacity_bytes,
                           avr_comm** out_comm) {
  return guarded([&]() -> int {
    require(out_comm != nullptr && name != nullptr && name[0] == '/', "invalid argument");
    *out_comm = nullptr;
    require(n_ranks >= 1 && n_ranks <= avr::SharedHeader::kMaxRanks && rank >= 0 && rank < n_ranks,
            "invalid rank");
    require(capacity_bytes > 0, "capacity_bytes must be positive");
    auto world = std::make_unique<avr::SharedWorld>();
    world->n_ranks = n_ranks;
    world->name = name;
    world->capacity = (capacity_bytes + 4095) / 4096 * 4096;
    world->owner = rank == 0;
    const size_t head = (sizeof(avr::SharedHeader) + 4095) / 4096 * 4096;
    world->mapped_bytes = head + world->capacity * static_cast<size_t>(n_ranks);
    // Rank 0 creates the segment afresh (a name left behind by a crashed job is removed first, and
    // O_EXCL makes sure nobody else's is adopted), sizes it -- a new segment is all zeros: the
    // barrier's state -- and publishes the magic word last; the others open WITHOUT creating,
    // retrying until the name exists, has its size and carries the magic.  The name is removed
    // right after the attach barrier below, so a rank that fails later leaks nothing in /dev/shm.
    const auto give_up = std::chrono::steady_clock::now() + std::chrono::seconds(120);
    auto pause = [&](const char* what) {
      if (std::chrono::steady_clock::now() > give_up) {
        throw std::runtime_error(std::string("shared communicator: ") + what + " within 120 s");
      }
      std::this_thread::sleep_for(std::chrono::microseconds(200));
    };
    int fd = -1;
    if (rank == 0) {
      (void)shm_unlink(name);
      fd = shm_open(name, O_CREAT | O_EXCL | O_RDWR, 0600);
      if (fd < 0) throw std::runtime_error(std::string("shm_open(") + name + ") failed");
      if (ftruncate(fd, static_cast<off_t>(world->mapped_bytes)) != 0) {
        (void)close(fd);
        (void)shm_unlink(name);
        throw std::runtime_error("ftruncate of the shared segment failed");
      }
    } else {
      for (;;) {
        fd = shm_open(name, O_RDWR, 0600);
        if (fd >= 0) {
          struct stat st {};
          if (fstat(fd, &st) == 0 && static_cast<size_t>(st.st_size) == world->mapped_bytes) break;
          (void)close(fd);
          fd = -1;
        }
        pause("rank 0 did not create the segment");
      }
    }
    void* mapping = mmap(nullptr, world->mapped_bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    (void)close(fd);
    if (mapping == MAP_FAILED) {
      if (rank == 0) (void)shm_unlink(name);
      throw std::runtime_error("mmap of the shared segment failed");
    }
    world->mapping = mapping;
    auto comm = std::make_unique<avr_comm>();
    comm->rank = rank;
    comm->n_ranks = n_ranks;
    comm->shared = std::move(world);
    avr::SharedHeader* header = comm->shared->header();
    if (rank == 0) {
      header->magic.store(avr::SharedHeader::kMagic, std::memory_order_release);
    } else {
      while (header->magic.load(std::memory_order_acquire) != avr::SharedHeader::kMagic) {
        pause("rank 0 did not initialise the segment");
      }
    }
    comm->shared->barrier(120000);  // everybody is attached before anybody communicates
    if (rank == 0) {                // ... and nobody needs the name any more
      (void)shm_unlink(name);
      comm->shared->owner = false;
    }
    *out_comm = comm.release();
    return AVR_OK;
  });
}

void avr_comm_destroy(avr_comm* comm) {
  if (comm == nullptr) return;
  if (comm->control_done != nullptr) (void)hipEventDestroy(comm->control_done);
  if (comm->control_dev != nullptr) (void)hipFree(comm->control_dev);
  if (comm->control_host != nullptr) (void)hipHostFree(comm->control_host);
  if (comm->nccl != nullptr) {
    try {
      (void)avr::rccl().comm_destroy(comm->nccl);
    } catch (...) {
    }
  }
  delete comm;
}

int avr_comm_set_control(avr_comm* comm, avr_control_allgather_fn allgather, void* user) {
  return guarded([&]() -> int {
    require(comm != nullptr, "null communicator");
    comm->control_fn = allgather;
    comm->control_user = user;
    return AVR_OK;
  });
}

long avr_comm_control_rounds(const avr_comm* comm) { return comm ? comm->control_rounds : -1; }

int avr_comm_control_allgather(avr_comm* comm, avr_context* ctx, const void* mine, void* all,
                               int bytes) {
  return guarded([&]() -> int {
    require(comm != nullptr && mine != nullptr && all != nullptr, "null argument");
    require(bytes > 0 && bytes <= AVR_CONTROL_MAX_BYTES && bytes % 4 == 0,
            "bytes must be a multiple of 4 in (0, AVR_CONTROL_MAX_BYTES]");
    const int n = comm->n_ranks, me = comm->rank;
    const size_t each = static_cast<size_t>(bytes);
    char* out = static_cast<char*>(all);
    ++comm->control_rounds;
    if (comm->control_fn != nullptr) {  // the caller's control plane
      // Under the same deadline as every other wait of a frame (AVR_FRAME_TIMEOUT_MS): plan
      // agreement and the co-run windows block the host inside avr_renderer_render, and a peer that
      // hangs would otherwise leave this rank in the caller's allgather for as long as THAT waits
      // (gloo: 30 minutes by default).  The callback runs on a helper thread over buffers the
      // helper co-owns; past the deadline the helper is left behind (it may still be inside the
      // caller's collective: the renderer is failed and the process is expected to exit).
      const int limit_ms = avr::frame_timeout_ms();
      if (limit_ms <= 0) {
        if (comm->control_fn(comm->control_user, mine, all, bytes) != 0) {
          throw std::runtime_error("control plane: the caller's allgather failed");
        }
        return AVR_OK;
      }
      struct Round {
        std::mutex mutex;
        std::condition_variable done_cv;
        bool done = false;
        int status = 0;
        std::vector<unsigned char> mine, all;
      };
      auto round = std::make_shared<Round>();
      round->mine.assign(static_cast<const unsigned char*>(mine), static_cast<const unsigned char*>(mine) + each);
      round->all.resize(each * static_cast<size_t>(n));
      const avr_control_allgather_fn fn = comm->control_fn;
      void* const user = comm->control_user;
      std::thread([round, fn, user, bytes] {
        int status = -1;
        try {
          status = fn(user, round->mine.data(), round->all.data(), bytes);
        } catch (...) {  // (a C callback must not throw; a C++ one that does is a failed round)
        }
        std::lock_guard<std::mutex> lock(round->mutex);
        round->status = status;
        round->done = true;
        round->done_cv.notify_all();
      }).detach();
      std::unique_lock<std::mutex> lock(round->mutex);
      if (!round->done_cv.wait_for(lock, std::chrono::milliseconds(limit_ms), [&] { return round->done; })) {
        throw avr::DeadlineExceeded("control plane: the caller's allgather did not finish within " +
                                    std::to_string(limit_ms) + " ms (AVR_FRAME_TIMEOUT_MS): a peer is "
                                    "missing, or the ranks' calls differ");
      }
      if (round->status != 0) throw std::runtime_error("control plane: the caller's allgather failed");
      std::memcpy(all, round->all.data(), round->all.size());
      return AVR_OK;
    }
    if (comm->solo) {  // alone: every peer "says" what this rank says
      for (int s = 0; s < n; ++s) std::memcpy(out + static_cast<size_t>(s) * each, mine, each);
      return AVR_OK;
    }
    if (comm->shared) {
      avr::SharedWorld& world = *comm->shared;
      std::memcpy(world.header()->control[me], mine, each);
      world.meet(me, 4, each);
      for (int s = 0; s < n; ++s) {
        std::memcpy(out + static_cast<size_t>(s) * each, world.header()->control[s], each);
      }
      world.barrier();  // everybody has read: the slots may be rewritten
      return AVR_OK;
    }
    if (comm->local) {
      avr::LocalWorld& world = *comm->local;
      {
        std::lock_guard<std::mutex> lock(world.mutex);
        world.control[static_cast<size_t>(me)].assign(static_cast<const unsigned char*>(mine),
                                                      static_cast<const unsigned char*>(mine) + each);
      }
      world.meet(me, 4, each);
      for (int s = 0; s < n; ++s) {
        if (world.control[static_cast<size_t>(s)].size() != each) {
          throw std::runtime_error("control plane: the ranks' calls differ (message sizes)");
        }
        std::memcpy(out + static_cast<size_t>(s) * each, world.control[static_cast<size_t>(s)].data(), each);
      }
      world.barrier();
      return AVR_OK;
    }
    // RCCL, in band: my slot to every peer, every peer's slot to me -- one grouped round of tiny
    // messages over the connections the frame's exchange uses anyway (no ring / tree is set up),
    // on the context's stream, behind whatever the frames before have queued there.  Every rank
    // calls this at the same point of the same frame, so the rounds meet in order.
    require(ctx != nullptr, "the in-band control plane needs a context (its stream)");
    hipStream_t stream = static_cast<hipStream_t>(avr::context_stream(ctx));
    const size_t slot = AVR_CONTROL_MAX_BYTES;
    if (comm->control_dev == nullptr) {
      const size_t total = slot * static_cast<size_t>(n + 1);
      avr::hip_ok(hipMalloc(&comm->control_dev, total), "hipMalloc(control)");
      avr::hip_ok(hipHostMalloc(&comm->control_host, total, hipHostMallocMapped), "hipHostMalloc(control)");
      avr::hip_ok(hipHostGetDevicePointer(&comm->control_host_mapped, comm->control_host, 0),
                  "hipHostGetDevicePointer(control)");
      avr::hip_ok(hipEventCreateWithFlags(&comm->control_done, hipEventDisableTiming), "hipEventCreate");
    }
    char* dev = static_cast<char*>(comm->control_dev);
    char* host = static_cast<char*>(comm->control_host);
    char* mapped = static_cast<char*>(comm->control_host_mapped);
    // slot 0: mine; slots 1 .. n: everybody's (mine copied there on the host, below)
    std::memcpy(host, mine, each);
    if (avr::launch_upload(mapped, dev, each, stream) != AVR_OK) throw std::runtime_error(avr_last_error());
    const avr::Rccl& api = avr::rccl();
    avr::RcclGroup group;
    for (int s = 0; s < n; ++s) {
      if (s == me && n > 1) continue;  // (a one-rank communicator talks to itself on purpose)
      avr::nccl_ok(api.send(dev, each, ncclChar, s, comm->nccl, stream), "ncclSend(control)");
      avr::nccl_ok(api.recv(dev + slot * static_cast<size_t>(s + 1), each, ncclChar, s, comm->nccl, stream),
                   "ncclRecv(control)");
    }
    group.end();
    if (avr::launch_upload(dev + slot, mapped + slot, slot * static_cast<size_t>(n), stream) != AVR_OK) {
      throw std::runtime_error(avr_last_error());
    }
    avr::hip_ok(hipEventRecord(comm->control_done, stream), "hipEventRecord(control)");
    avr::wait_event_deadline(comm->control_done,
                             "control plane: the in-band round (a peer is missing, or the ranks' "
                             "calls differ)");
    for (int s = 0; s < n; ++s) {
      std::memcpy(out + static_cast<size_t>(s) * each, host + slot * static_cast<size_t>(s + 1), each);
    }
    if (n > 1) std::memcpy(out + static_cast<size_t>(me) * each, mine, each);
    return AVR_OK;
  });
}

int avr_frame_plan_agree(const avr_frame_plan* plan, avr_comm* comm, avr_context* ctx,
                         uint64_t settings_digest) {
  return guarded([&]() -> int {
    check_plan(comm, plan);
    const int n = comm->n_ranks;
    const size_t un = static_cast<size_t>(n);
    const int bytes = static_cast<int>(8 + 16 * un);
    require(bytes <= AVR_CONTROL_MAX_BYTES, "too many ranks for the plan agreement check");
    if (plan->send_splits.size() != un || plan->recv_splits.size() != un) {
      throw std::runtime_error("frame plan: split tables do not match the rank count");
    }
    // FNV-1a over everything of the plan that is the same on every rank by construction
    uint64_t digest = 0xcbf29ce484222325ull;
    auto mix = [&](const void* data, size_t size) {
      const unsigned char* p = static_cast<const unsigned char*>(data);
      for (size_t i = 0; i < size; ++i) {
        digest ^= p[i];
        digest *= 0x100000001b3ull;
      }
    };
    auto mix_value = [&](auto value) { mix(&value, sizeof(value)); };
    mix_value(settings_digest);
    mix_value(plan->info.n_ranks);
    mix_value(plan->info.n_runs_total);
    mix_value(plan->info.n_pixels);
    mix_value(plan->pieces.layout);
    mix_value(plan->pieces.band_rows);
    mix_value(plan->pieces.n_pieces);
    mix_value(plan->pieces.width);
    mix_value(plan->pieces.height);
    mix_value(plan->pieces.piece_size);
    mix_value(static_cast<int32_t>(plan->tightened ? 1 : 0));
    mix(plan->group_order.data(), plan->group_order.size() * sizeof(int32_t));
    mix(plan->piece_of_rank.data(), plan->piece_of_rank.size() * sizeof(int32_t));
    mix(plan->layer_box.data(), plan->layer_box.size() * sizeof(int32_t));
    for (const avr_run_info& run : plan->runs) {
      mix_value(run.owner);
      mix_value(run.local_run);
      mix_value(run.first_layer);
      mix_value(run.n_layers);
      mix(run.rect, sizeof(run.rect));
    }
    std::vector<int64_t> mine(1 + 2 * un), all((1 + 2 * un) * un);
    std::memcpy(&mine[0], &digest, 8);
    std::copy(plan->send_splits.begin(), plan->send_splits.end(), mine.begin() + 1);
    std::copy(plan->recv_splits.begin(), plan->recv_splits.end(), mine.begin() + 1 + n);
    const int status = avr_comm_control_allgather(comm, ctx, mine.data(), all.data(), bytes);
    if (status != AVR_OK) return status;
    plan->agreed_digest = digest;
    if (comm->solo) return AVR_OK;  // (alone: nobody to agree with)
    const size_t row = 1 + 2 * un;
    for (size_t a = 1; a < un; ++a) {  // every rank scans the same matrix: the same verdict everywhere
      if (all[a * row] != all[0]) {
        throw std::runtime_error("frame plan: rank " + std::to_string(a) + "'s plan differs from rank 0's "
                                 "(image, pieces, group order, runs or settings): the ranks were not "
                                 "given the same frame");
      }
    }
    for (size_t a = 0; a < un; ++a) {
      for (size_t b = 0; b < un; ++b) {
        const int64_t sends = all[a * row + 1 + b], expects = all[b * row + 1 + un + a];
        if (sends != expects) {
          throw std::runtime_error("frame plan: rank " + std::to_string(a) + " sends " +
                                   std::to_string(sends) + " floats to rank " + std::to_string(b) +
                                   ", which expects " + std::to_string(expects) +
                                   ": the exchange would never complete");
        }
      }
    }
    return AVR_OK;
  });
}

int avr_comm_rank(const avr_comm* comm) { return comm ? comm->rank : -1; }
int avr_comm_size(const avr_comm* comm) { return comm ? comm->n_ranks : -1; }

namespace {
int exchange(avr_context* ctx, const avr_frame_plan* plan, avr_comm* comm, const float* send,
             float* recv, bool move_own, const avr_gather_op* rider = nullptr);
void run_gather(avr_context* ctx, avr_comm* comm, const avr_gather_op& op, bool in_group);
}

int avr_exchange(avr_context* ctx, const avr_frame_plan* plan, avr_comm* comm, const float* send,
                 float* recv) {
  return exchange(ctx, plan, comm, send, recv, /*move_own=*/true);
}

int avr_exchange_peers(avr_context* ctx, const avr_frame_plan* plan, avr_comm* comm,
                       const float* send, float* recv) {
  return exchange(ctx, plan, comm, send, recv, /*move_own=*/false);
}

int avr_exchange_peers_gather(avr_context* ctx, const avr_frame_plan* plan, avr_comm* comm,
                              const float* send, float* recv, const avr_gather_op* rider) {
  return exchange(ctx, plan, comm, send, recv, /*move_own=*/false, rider);
}

}  // extern "C"

namespace {
// move_own: whether the rank's block for itself is copied from the send to the receive buffer
// (avr_fold_plan_own reads it where it is)
// rider: a gather (of an earlier frame's pieces) that travels in the same grouped RCCL round -- one
// launch per frame on the compositing stream instead of two
int exchange(avr_context* ctx, const avr_frame_plan* plan, avr_comm* comm, const float* send,
             float* recv, bool move_own, const avr_gather_op* rider) {
  return guarded([&]() -> int {
    check_plan(comm, plan);
    hipStream_t stream = static_cast<hipStream_t>(avr::context_stream(ctx));
    const int n = comm->n_ranks, me = comm->rank;
    require(plan->info.send_floats == 0 || send != nullptr, "null send buffer");
    require(plan->info.recv_floats == 0 || recv != nullptr, "null receive buffer");
    std::vector<int64_t> send_at(static_cast<size_t>(n) + 1, 0), recv_at(static_cast<size_t>(n) + 1, 0);
    for (int s = 0; s < n; ++s) {
      send_at[static_cast<size_t>(s) + 1] = send_at[static_cast<size_t>(s)] + plan->send_splits[static_cast<size_t>(s)];
      recv_at[static_cast<size_t>(s) + 1] = recv_at[static_cast<size_t>(s)] + plan->recv_splits[static_cast<size_t>(s)];
    }
    if (plan->send_splits[static_cast<size_t>(me)] != plan->recv_splits[static_cast<size_t>(me)]) {
      throw std::runtime_error("frame plan: a rank's block for itself differs between send and receive layout");
    }
    if (comm->solo) {  // only the block the rank keeps for itself moves
      const int64_t own = plan->send_splits[static_cast<size_t>(me)];
      if (own > 0 && move_own) {
        avr::hip_ok(hipMemcpyAsync(recv + recv_at[static_cast<size_t>(me)], send + send_at[static_cast<size_t>(me)],
                                   static_cast<size_t>(own) * 4, hipMemcpyDeviceToDevice, stream),
                    "hipMemcpyAsync(exchange)");
      }
      if (comm->nccl != nullptr) {
        // ... and the blocks for the peers go through RCCL to THIS rank (a one-rank communicator):
        // the grouped send / receive round of the real frame -- its launch on the host, its
        // kernel beside the paint kernels, its bytes through HBM -- with the links left out.
        // Block s is sent and received with the smaller of its two sizes -- or the given share
        // of it: on the real node the seven blocks travel over seven links at once, here one
        // after the other over one connection (timing only: what lands in the receive buffer is
        // this rank's own data).
        // (percent 0, "one link": only the LARGEST peer block, whole -- RCCL works off the operations
        // for one peer one after the other, ~13 us each, so N - 1 pairs to the rank itself measure
        // that serialisation, which the node, with one peer per link, does not have)
        const avr::Rccl& api = avr::rccl();
        int busiest = -1;
        if (comm->solo_percent == 0) {
          int64_t most = 0;
          for (int s = 0; s < n; ++s) {
            const int64_t count = (s == me) ? 0 : std::min(plan->send_splits[static_cast<size_t>(s)],
                                                           plan->recv_splits[static_cast<size_t>(s)]);
            if (count > most) {
              most = count;
              busiest = s;
            }
          }
        }
        avr::RcclGroup group;
        for (int s = 0; s < n; ++s) {
          if (s == me) continue;
          if (comm->solo_percent == 0 && s != busiest) continue;
          const int64_t count = std::min(plan->send_splits[static_cast<size_t>(s)],
                                         plan->recv_splits[static_cast<size_t>(s)]) *
                                (comm->solo_percent == 0 ? 100 : comm->solo_percent) / 100;
          if (count <= 0) continue;
          avr::nccl_ok(api.send(send + send_at[static_cast<size_t>(s)], static_cast<size_t>(count), ncclFloat, 0,
                                comm->nccl, stream), "ncclSend");
          avr::nccl_ok(api.recv(recv + recv_at[static_cast<size_t>(s)], static_cast<size_t>(count), ncclFloat, 0,
                                comm->nccl, stream), "ncclRecv");
        }
        if (rider != nullptr) run_gather(ctx, comm, *rider, /*in_group=*/true);
        group.end();
      } else if (rider != nullptr) {
        run_gather(ctx, comm, *rider, /*in_group=*/false);
      }
      return AVR_OK;
    }
    if (comm->shared) {
      // cross-process rehearsal: my whole send buffer into my region, meet, pull my blocks, meet
      avr::SharedWorld& world = *comm->shared;
      const int64_t total = send_at[static_cast<size_t>(n)] * 4;
      if (static_cast<size_t>(total) > world.capacity) {
        throw std::runtime_error("shared communicator: the send buffer exceeds the segment's capacity");
      }
      std::exception_ptr failure;
      try {
        drain(stream);
        if (total > 0) {
          avr::hip_ok(hipMemcpy(world.region(me), send, static_cast<size_t>(total), hipMemcpyDeviceToHost),
                      "hipMemcpy(exchange)");
        }
        for (int s = 0; s < n; ++s) {
          world.header()->offsets[me][s] = send_at[static_cast<size_t>(s)] * 4;
          world.header()->sizes[me][s] = plan->send_splits[static_cast<size_t>(s)] * 4;
        }
      } catch (...) {
        failure = std::current_exception();
      }
      try {
        world.meet(me, 1, 0);
      } catch (const avr::CallsDiffer&) {
        throw;
      } catch (const avr::DeadlineExceeded&) {
        throw;
      } catch (...) {
        if (!failure) failure = std::current_exception();
      }
      try {
        if (failure) std::rethrow_exception(failure);
        for (int s = 0; s < n; ++s) {
          const int64_t bytes = world.header()->sizes[s][me];
          if (bytes != plan->recv_splits[static_cast<size_t>(s)] * 4) {
            throw std::runtime_error("exchange: the ranks' frame plans disagree on a block size");
          }
          if (bytes == 0 || (s == me && !move_own)) continue;
          avr::hip_ok(hipMemcpy(recv + recv_at[static_cast<size_t>(s)],
                                world.region(s) + world.header()->offsets[s][me],
                                static_cast<size_t>(bytes), hipMemcpyHostToDevice),
                      "hipMemcpy(exchange)");
        }
      } catch (...) {
        failure = std::current_exception();
      }
      world.barrier();  // every rank has pulled: the regions may be rewritten
      if (failure) std::rethrow_exception(failure);
      if (rider != nullptr) run_gather(ctx, comm, *rider, /*in_group=*/false);
      return AVR_OK;
    }
    if (comm->local) {
      // in-process rehearsal: publish, meet, pull, meet
      avr::LocalWorld& world = *comm->local;
      drain(stream);  // my send buffer is complete
      world.base[static_cast<size_t>(me)] = reinterpret_cast<const char*>(send);
      world.offsets[static_cast<size_t>(me)].assign(static_cast<size_t>(n), 0);
      world.sizes[static_cast<size_t>(me)].assign(static_cast<size_t>(n), 0);
      for (int s = 0; s < n; ++s) {
        world.offsets[static_cast<size_t>(me)][static_cast<size_t>(s)] = send_at[static_cast<size_t>(s)] * 4;
        world.sizes[static_cast<size_t>(me)][static_cast<size_t>(s)] = plan->send_splits[static_cast<size_t>(s)] * 4;
      }
      // (whatever goes wrong between the two meetings, the second one is still attended: the
      // peers would otherwise wait for this rank forever)
      std::exception_ptr failure;
      try {
        world.meet(me, 1, 0);
        for (int s = 0; s < n; ++s) {
          const int64_t bytes = world.sizes[static_cast<size_t>(s)][static_cast<size_t>(me)];
          if (bytes != plan->recv_splits[static_cast<size_t>(s)] * 4) {
            throw std::runtime_error("exchange: the ranks' frame plans disagree on a block size");
          }
          if (bytes == 0 || (s == me && !move_own)) continue;
          avr::hip_ok(hipMemcpyAsync(recv + recv_at[static_cast<size_t>(s)],
                                     world.base[static_cast<size_t>(s)] +
                                         world.offsets[static_cast<size_t>(s)][static_cast<size_t>(me)],
                                     static_cast<size_t>(bytes), hipMemcpyDeviceToDevice, stream),
                      "hipMemcpyAsync(exchange)");
        }
        drain(stream);
      } catch (const avr::CallsDiffer&) {
        throw;
      } catch (const avr::DeadlineExceeded&) {
        throw;
      } catch (...) {
        failure = std::current_exception();
      }
      world.barrier();  // every rank has pulled: the send buffers may be rewritten
      if (failure) std::rethrow_exception(failure);
      if (rider != nullptr) run_gather(ctx, comm, *rider, /*in_group=*/false);
      return AVR_OK;
    }
    // RCCL: one grouped round; a rank's block for itself is a device copy
    const avr::Rccl& api = avr::rccl();
    const int64_t own = plan->send_splits[static_cast<size_t>(me)];
    if (n == 1) {
      // a one-rank communicator only exists to exercise this path where a single GPU is all
      // there is: the block for itself goes through ncclSend / ncclRecv like any other
      if ((own > 0 && move_own) || rider != nullptr) {
        avr::RcclGroup group;
        if (own > 0 && move_own) {
          avr::nccl_ok(api.send(send, static_cast<size_t>(own), ncclFloat, 0, comm->nccl, stream), "ncclSend");
          avr::nccl_ok(api.recv(recv, static_cast<size_t>(own), ncclFloat, 0, comm->nccl, stream), "ncclRecv");
        }
        if (rider != nullptr) run_gather(ctx, comm, *rider, /*in_group=*/true);
        group.end();
      }
      return AVR_OK;
    }
    if (own > 0 && move_own) {
      avr::hip_ok(hipMemcpyAsync(recv + recv_at[static_cast<size_t>(me)], send + send_at[static_cast<size_t>(me)],
                                 static_cast<size_t>(own) * 4, hipMemcpyDeviceToDevice, stream),
                  "hipMemcpyAsync(exchange)");
    }
    avr::RcclGroup group;
    for (int s = 0; s < n; ++s) {
      if (s == me) continue;
      const int64_t out = plan->send_splits[static_cast<size_t>(s)];
      const int64_t in = plan->recv_splits[static_cast<size_t>(s)];
      if (out > 0) {
        avr::nccl_ok(api.send(send + send_at[static_cast<size_t>(s)], static_cast<size_t>(out), ncclFloat, s,
                              comm->nccl, stream), "ncclSend");
      }
      if (in > 0) {
        avr::nccl_ok(api.recv(recv + recv_at[static_cast<size_t>(s)], static_cast<size_t>(in), ncclFloat, s,
                              comm->nccl, stream), "ncclRecv");
      }
    }
    if (rider != nullptr) run_gather(ctx, comm, *rider, /*in_group=*/true);
    group.end();
    return AVR_OK;
  });
}
}  // namespace

extern "C" {

int avr_exchange_pieces(avr_context* ctx, avr_comm* comm, const int32_t* group_order,
                        int64_t n_pixels, int bytes_per_pixel, const void* image, void* slices) {
  return guarded([&]() -> int {
    require(comm != nullptr, "null communicator");
    hipStream_t stream = static_cast<hipStream_t>(avr::context_stream(ctx));
    const int n = comm->n_ranks, me = comm->rank;
    require(n_pixels >= 0 && bytes_per_pixel > 0, "invalid image description");
    // group position k <-> rank
    std::vector<int> rank_at(static_cast<size_t>(n)), position_of(static_cast<size_t>(n), -1);
    for (int k = 0; k < n; ++k) {
      const int member = group_order != nullptr ? group_order[k] : k;
      require(member >= 0 && member < n && position_of[static_cast<size_t>(member)] < 0,
              "group_order must be a permutation of the ranks");
      rank_at[static_cast<size_t>(k)] = member;
      position_of[static_cast<size_t>(member)] = k;
    }
    const int64_t piece_size = n_pixels / n;  // getPieceRange (DirectSendBase.cpp:59-74)
    auto piece_range = [&](int k, int64_t* begin, int64_t* end) {
      *begin = piece_size * k;
      *end = (k < n - 1) ? *begin + piece_size : n_pixels;
    };
    int64_t my_begin = 0, my_end = 0;
    const int my_position = position_of[static_cast<size_t>(me)];
    piece_range(my_position, &my_begin, &my_end);
    const int64_t my_bytes = (my_end - my_begin) * bytes_per_pixel;
    require(n_pixels == 0 || (image != nullptr && slices != nullptr), "null image");
    const char* src = static_cast<const char*>(image);
    char* dst = static_cast<char*>(slices);
    // the block this rank keeps: its own piece of its own image, at its own group position
    auto keep_own = [&] {
      if (my_bytes > 0) {
        avr::hip_ok(hipMemcpyAsync(dst + my_position * my_bytes, src + my_begin * bytes_per_pixel,
                                   static_cast<size_t>(my_bytes), hipMemcpyDeviceToDevice, stream),
                    "hipMemcpyAsync(exchange_pieces)");
      }
    };
    if (comm->solo || (n == 1 && comm->nccl == nullptr && !comm->local)) {
      keep_own();
      return AVR_OK;
    }
    if (comm->shared) {
      avr::SharedWorld& world = *comm->shared;
      const size_t image_bytes = static_cast<size_t>(n_pixels) * static_cast<size_t>(bytes_per_pixel);
      if (image_bytes > world.capacity) {
        throw std::runtime_error("shared communicator: the image exceeds the segment's capacity");
      }
      std::exception_ptr failure;
      try {
        drain(stream);
        if (image_bytes > 0) {
          avr::hip_ok(hipMemcpy(world.region(me), src, image_bytes, hipMemcpyDeviceToHost),
                      "hipMemcpy(exchange_pieces)");
        }
      } catch (...) {
        failure = std::current_exception();
      }
      try {
        world.meet(me, 2, static_cast<uint64_t>(n_pixels) * static_cast<uint64_t>(bytes_per_pixel));
      } catch (const avr::CallsDiffer&) {
        throw;
      } catch (const avr::DeadlineExceeded&) {
        throw;
      } catch (...) {
        if (!failure) failure = std::current_exception();
      }
      try {
        if (failure) std::rethrow_exception(failure);
        for (int k = 0; k < n && my_bytes > 0; ++k) {
          avr::hip_ok(hipMemcpy(dst + k * my_bytes,
                                world.region(rank_at[static_cast<size_t>(k)]) + my_begin * bytes_per_pixel,
                                static_cast<size_t>(my_bytes), hipMemcpyHostToDevice),
                      "hipMemcpy(exchange_pieces)");
        }
      } catch (...) {
        failure = std::current_exception();
      }
      world.barrier();
      if (failure) std::rethrow_exception(failure);
      return AVR_OK;
    }
    if (comm->local) {
      avr::LocalWorld& world = *comm->local;
      drain(stream);  // my image is complete
      world.base[static_cast<size_t>(me)] = src;
      std::exception_ptr failure;
      try {
        world.meet(me, 2, static_cast<uint64_t>(n_pixels) * static_cast<uint64_t>(bytes_per_pixel));
        for (int k = 0; k < n; ++k) {  // the image of the rank at position k, my piece of it
          if (my_bytes == 0) break;
          avr::hip_ok(hipMemcpyAsync(dst + k * my_bytes,
                                     world.base[static_cast<size_t>(rank_at[static_cast<size_t>(k)])] +
                                         my_begin * bytes_per_pixel,
                                     static_cast<size_t>(my_bytes), hipMemcpyDeviceToDevice, stream),
                      "hipMemcpyAsync(exchange_pieces)");
        }
        drain(stream);
      } catch (const avr::CallsDiffer&) {
        throw;
      } catch (const avr::DeadlineExceeded&) {
        throw;
      } catch (...) {
        failure = std::current_exception();
      }
      world.barrier();  // every rank has pulled: the images may be rewritten
      if (failure) std::rethrow_exception(failure);
      return AVR_OK;
    }
    const avr::Rccl& api = avr::rccl();
    if (n > 1) keep_own();
    avr::RcclGroup group;
    for (int k = 0; k < n; ++k) {
      const int peer = rank_at[static_cast<size_t>(k)];
      if (peer == me && n > 1) continue;  // (a one-rank communicator sends to itself on purpose)
      int64_t b = 0, e = 0;
      piece_range(k, &b, &e);
      if (e > b) {  // piece k of my image -> the rank at position k
        avr::nccl_ok(api.send(src + b * bytes_per_pixel, static_cast<size_t>((e - b) * bytes_per_pixel),
                              ncclChar, peer, comm->nccl, stream), "ncclSend");
      }
      if (my_bytes > 0) {  // my piece of its image, into the block of its position
        avr::nccl_ok(api.recv(dst + k * my_bytes, static_cast<size_t>(my_bytes), ncclChar, peer,
                              comm->nccl, stream), "ncclRecv");
      }
    }
    group.end();
    return AVR_OK;
  });
}

}  // extern "C"

namespace {
// ImageFull::Gather of pieces whose places in the gathered buffer are given as pixel ranges (of
// the frame's plan -- possibly an EARLIER frame's: the frame driver lets the gather of frame f ride
// in the grouped round of frame f + 1).  in_group: an RCCL group is open (the caller closes it);
// skip_own: the root does not copy its own piece (its assemble pass reads it where it is).
void run_gather(avr_context* ctx, avr_comm* comm, const avr_gather_op& op, bool in_group) {
  hipStream_t stream = static_cast<hipStream_t>(avr::context_stream(ctx));
  const int n = comm->n_ranks, me = comm->rank;
  const int bytes_per_pixel = op.bytes_per_pixel, root = op.root;
  const void* piece = op.piece;
  void* full = op.full;
  require(bytes_per_pixel > 0 && root >= 0 && root < n && op.begin != nullptr && op.end != nullptr,
          "invalid gather");
  auto piece_range = [&](int rank, int64_t* begin, int64_t* end) {
    *begin = op.begin[rank];
    *end = op.end[rank];
  };
  const bool copy_own = op.skip_own == 0;
  int64_t my_begin = 0, my_end = 0;
  piece_range(me, &my_begin, &my_end);
  require(my_end == my_begin || piece != nullptr, "null piece");
  bool anything = false;
  for (int s = 0; s < n; ++s) anything = anything || op.end[s] > op.begin[s];
  require(me != root || full != nullptr || !anything, "null destination on the root");
  char* dst = static_cast<char*>(full);
  if (comm->solo) {
    if (me == root && my_end > my_begin && copy_own) {
      avr::hip_ok(hipMemcpyAsync(dst + my_begin * bytes_per_pixel, piece,
                                 static_cast<size_t>(my_end - my_begin) * bytes_per_pixel,
                                 hipMemcpyDeviceToDevice, stream), "hipMemcpyAsync(gather)");
    }
    if (comm->nccl != nullptr && my_end > my_begin) {
      // through RCCL to this rank itself (timing only, as avr_exchange): the root receives a
      // piece from every other rank, the others send theirs
      const avr::Rccl& api = avr::rccl();
      const size_t bytes = static_cast<size_t>(my_end - my_begin) * static_cast<size_t>(bytes_per_pixel);
      avr::RcclGroup group(!in_group);
      if (me != root) {
        // (no destination buffer off the root: the piece's first half lands on its second --
        // the piece has been handed over by then, and a solo rank's pixels mean nothing)
        const size_t half = bytes / 2;
        if (half > 0) {
          char* target = static_cast<char*>(const_cast<void*>(piece)) + half;
          avr::nccl_ok(api.send(piece, half, ncclUint8, 0, comm->nccl, stream), "ncclSend");
          avr::nccl_ok(api.recv(target, half, ncclUint8, 0, comm->nccl, stream), "ncclRecv");
        }
      } else {
        bool one_done = false;
        for (int s = 0; s < n; ++s) {
          if (s == me) continue;
          if (comm->solo_percent == 0 && one_done) break;  // "one link": one peer's piece
          int64_t b = 0, e = 0;
          piece_range(s, &b, &e);
          const size_t count = std::min(bytes, static_cast<size_t>(e - b) * static_cast<size_t>(bytes_per_pixel));
          if (count == 0) continue;
          one_done = true;
          avr::nccl_ok(api.send(piece, count, ncclUint8, 0, comm->nccl, stream), "ncclSend");
          avr::nccl_ok(api.recv(dst + b * bytes_per_pixel, count, ncclUint8, 0, comm->nccl, stream),
                       "ncclRecv");
        }
      }
      group.end();
    }
    return;
  }
  if (comm->shared) {
    avr::SharedWorld& world = *comm->shared;
    const size_t mine = static_cast<size_t>(my_end - my_begin) * static_cast<size_t>(bytes_per_pixel);
    if (mine > world.capacity) {
      throw std::runtime_error("shared communicator: the piece exceeds the segment's capacity");
    }
    std::exception_ptr failure;
    try {
      drain(stream);
      if (mine > 0 && me != root) {
        avr::hip_ok(hipMemcpy(world.region(me), piece, mine, hipMemcpyDeviceToHost), "hipMemcpy(gather)");
      }
    } catch (...) {
      failure = std::current_exception();
    }
    try {
      world.meet(me, 3, static_cast<uint64_t>(bytes_per_pixel));
    } catch (const avr::CallsDiffer&) {
      throw;
    } catch (const avr::DeadlineExceeded&) {
      throw;
    } catch (...) {
      if (!failure) failure = std::current_exception();
    }
    try {
      if (failure) std::rethrow_exception(failure);
      if (me == root) {
        for (int s = 0; s < n; ++s) {
          int64_t b = 0, e = 0;
          piece_range(s, &b, &e);
          if (e == b) continue;
          const size_t bytes = static_cast<size_t>(e - b) * static_cast<size_t>(bytes_per_pixel);
          if (s == me) {
            if (copy_own) {
              // (a device-to-device hipMemcpy may return before it has run, and the context's
              // stream is non-blocking: what follows on it would not wait for the null stream)
              avr::hip_ok(hipMemcpy(dst + b * bytes_per_pixel, piece, bytes, hipMemcpyDeviceToDevice),
                          "hipMemcpy(gather)");
              avr::hip_ok(hipStreamSynchronize(nullptr), "hipStreamSynchronize(gather)");
            }
          } else {
            avr::hip_ok(hipMemcpy(dst + b * bytes_per_pixel, world.region(s), bytes, hipMemcpyHostToDevice),
                        "hipMemcpy(gather)");
          }
        }
      }
    } catch (...) {
      failure = std::current_exception();
    }
    world.barrier();
    if (failure) std::rethrow_exception(failure);
    return;
  }
  if (comm->local) {
    avr::LocalWorld& world = *comm->local;
    drain(stream);
    world.base[static_cast<size_t>(me)] = static_cast<const char*>(piece);
    std::exception_ptr failure;
    try {
      world.meet(me, 3, static_cast<uint64_t>(bytes_per_pixel));
      if (me == root) {
        for (int s = 0; s < n; ++s) {
          int64_t b = 0, e = 0;
          piece_range(s, &b, &e);
          if (e == b || (s == me && !copy_own)) continue;
          avr::hip_ok(hipMemcpyAsync(dst + b * bytes_per_pixel, world.base[static_cast<size_t>(s)],
                                     static_cast<size_t>(e - b) * bytes_per_pixel,
                                     hipMemcpyDeviceToDevice, stream), "hipMemcpyAsync(gather)");
        }
        drain(stream);
      }
    } catch (const avr::CallsDiffer&) {
      throw;
    } catch (const avr::DeadlineExceeded&) {
      throw;
    } catch (...) {
      failure = std::current_exception();
    }
    world.barrier();
    if (failure) std::rethrow_exception(failure);
    return;
  }
  const avr::Rccl& api = avr::rccl();
  if (n == 1) {  // as in avr_exchange: the one-rank case goes through RCCL on purpose
    if (my_end > my_begin) {
      const size_t bytes = static_cast<size_t>(my_end - my_begin) * bytes_per_pixel;
      avr::RcclGroup group(!in_group);
      avr::nccl_ok(api.send(piece, bytes, ncclChar, 0, comm->nccl, stream), "ncclSend");
      avr::nccl_ok(api.recv(dst + my_begin * bytes_per_pixel, bytes, ncclChar, 0, comm->nccl, stream),
                   "ncclRecv");
      group.end();
    }
    return;
  }
  if (me == root && my_end > my_begin && copy_own) {
    avr::hip_ok(hipMemcpyAsync(dst + my_begin * bytes_per_pixel, piece,
                               static_cast<size_t>(my_end - my_begin) * bytes_per_pixel,
                               hipMemcpyDeviceToDevice, stream), "hipMemcpyAsync(gather)");
  }
  avr::RcclGroup group(!in_group);
  if (me == root) {
    for (int s = 0; s < n; ++s) {
      if (s == root) continue;
      int64_t b = 0, e = 0;
      piece_range(s, &b, &e);
      if (e == b) continue;
      avr::nccl_ok(api.recv(dst + b * bytes_per_pixel, static_cast<size_t>(e - b) * bytes_per_pixel,
                            ncclChar, s, comm->nccl, stream), "ncclRecv");
    }
  } else if (my_end > my_begin) {
    avr::nccl_ok(api.send(piece, static_cast<size_t>(my_end - my_begin) * bytes_per_pixel, ncclChar,
                          root, comm->nccl, stream), "ncclSend");
  }
  group.end();
}

// the ranges of a plan's pieces in the gathered buffer: a rank's pixel range of the image
// (getPieceRange, DirectSendBase.cpp:59-74) or, with row bands, piece after piece
// (avr_assemble_rows restores the image order)
void plan_piece_ranges(const avr_frame_plan* plan, std::vector<int64_t>* begin, std::vector<int64_t>* end) {
  const size_t n = static_cast<size_t>(plan->info.n_ranks);
  begin->assign(n, 0);
  end->assign(n, 0);
  for (size_t rank = 0; rank < n; ++rank) {
    avr::piece_pixel_range(plan->pieces, plan->piece_of_rank[rank], &(*begin)[rank], &(*end)[rank]);
  }
}
}  // namespace

extern "C" {

int avr_gather(avr_context* ctx, const avr_frame_plan* plan, avr_comm* comm, const void* piece,
               int bytes_per_pixel, void* full, int root) {
  return guarded([&]() -> int {
    check_plan(comm, plan);
    std::vector<int64_t> begin, end;
    plan_piece_ranges(plan, &begin, &end);
    avr_gather_op op{};
    op.piece = piece;
    op.bytes_per_pixel = bytes_per_pixel;
    op.root = root;
    op.full = full;
    op.begin = begin.data();
    op.end = end.data();
    op.skip_own = 0;
    run_gather(ctx, comm, op, /*in_group=*/false);
    return AVR_OK;
  });
}

int avr_frame_plan_piece_ranges(const avr_frame_plan* plan, int64_t* begin_out, int64_t* end_out) {
  return guarded([&]() -> int {
    require(plan != nullptr && begin_out != nullptr && end_out != nullptr, "null argument");
    std::vector<int64_t> begin, end;
    plan_piece_ranges(plan, &begin, &end);
    std::copy(begin.begin(), begin.end(), begin_out);
    std::copy(end.begin(), end.end(), end_out);
    return AVR_OK;
  });
}

int avr_gather_run(avr_context* ctx, avr_comm* comm, const avr_gather_op* op) {
  return guarded([&]() -> int {
    require(comm != nullptr && op != nullptr, "null argument");
    run_gather(ctx, comm, *op, /*in_group=*/false);
    return AVR_OK;
  });
}


}  // extern "C"
