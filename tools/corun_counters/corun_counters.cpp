// Device-wide counter sampling while the frame pipeline runs: what the per-dispatch PMC passes
// (rocprofv3 --pmc, tools/pmc_passes.sh) cannot show, because they serialise the kernels -- the
// counters of the classify pass and the march SIDE BY SIDE.
//
// A rocprofiler-sdk tool library (rocprofiler_configure), loaded into the process by
//   ROCP_TOOL_LIBRARIES=$PWD/tools/corun_counters/libcorun_counters.so python3 tools/corun_counters/steady.py ...
// It opens the device counting service of GPU 0 and, on its own thread, times windows of
// AVR_COUNTER_WINDOW_MS (default 250): start the context with one group of counters, sleep, read
// the accumulated values, stop -- group after group (AVR_COUNTER_GROUPS: names separated by
// spaces, groups by ';'), until the process ends.  One line per counter and window goes to
// AVR_COUNTER_LOG (default corun_counters.log):
//   <window begin, CLOCK_MONOTONIC s> <window length s> <counter> <sum over all its instances>
// tools/corun_counters/steady.py writes the begin and end of its steady phase with the same clock,
// tools/corun_counters/summarise.py keeps the windows that lie inside it and prints per-frame values.
#include <rocprofiler-sdk/registration.h>
#include <rocprofiler-sdk/rocprofiler.h>

#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <ctime>
#include <map>
#include <sstream>
#include <string>
#include <thread>
#include <vector>

namespace {

struct Group {
  std::vector<std::string> names;
  rocprofiler_counter_config_id_t config{};
  size_t records = 0;
  bool usable = false;
};

rocprofiler_client_id_t* g_client = nullptr;
rocprofiler_client_finalize_t g_finalize = nullptr;
rocprofiler_context_id_t g_context{};
rocprofiler_buffer_id_t g_buffer{};
rocprofiler_agent_id_t g_agent{};
rocprofiler_counter_config_id_t g_current{};
std::vector<Group> g_groups;
std::map<uint64_t, std::string> g_names;  // counter id -> name
std::thread* g_thread = nullptr;
std::atomic<bool> g_stop{false}, g_stopped{false};
FILE* g_log = nullptr;

bool ok(rocprofiler_status_t status, const char* what) {
  if (status == ROCPROFILER_STATUS_SUCCESS) return true;
  std::fprintf(stderr, "corun_counters: %s: %s\n", what, rocprofiler_get_status_string(status));
  return false;
}

double now_s() {
  timespec t{};
  clock_gettime(CLOCK_MONOTONIC, &t);
  return static_cast<double>(t.tv_sec) + 1e-9 * static_cast<double>(t.tv_nsec);
}

bool first_gpu(rocprofiler_agent_id_t* out) {
  struct Found {
    rocprofiler_agent_id_t id{};
    bool any = false;
  } found;
  const auto each = [](rocprofiler_agent_version_t, const void** agents, size_t count, void* data) {
    auto* f = static_cast<Found*>(data);
    for (size_t i = 0; i < count && !f->any; ++i) {
      const auto* agent = static_cast<const rocprofiler_agent_v0_t*>(agents[i]);
      if (agent->type == ROCPROFILER_AGENT_TYPE_GPU) {
        f->id = agent->id;
        f->any = true;
      }
    }
    return ROCPROFILER_STATUS_SUCCESS;
  };
  if (!ok(rocprofiler_query_available_agents(ROCPROFILER_AGENT_INFO_VERSION_0, each,
                                             sizeof(rocprofiler_agent_t), &found),
          "query agents")) {
    return false;
  }
  *out = found.id;
  return found.any;
}

// name -> id of every counter the agent offers, and how many records (instances) each yields
bool supported(std::map<std::string, rocprofiler_counter_id_t>* by_name,
               std::map<uint64_t, size_t>* instances) {
  std::vector<rocprofiler_counter_id_t> ids;
  const auto each = [](rocprofiler_agent_id_t, rocprofiler_counter_id_t* counters, size_t count, void* data) {
    auto* v = static_cast<std::vector<rocprofiler_counter_id_t>*>(data);
    v->insert(v->end(), counters, counters + count);
    return ROCPROFILER_STATUS_SUCCESS;
  };
  if (!ok(rocprofiler_iterate_agent_supported_counters(g_agent, each, &ids), "list counters")) return false;
  for (const rocprofiler_counter_id_t id : ids) {
    rocprofiler_counter_info_v1_t info{};
    if (rocprofiler_query_counter_info(id, ROCPROFILER_COUNTER_INFO_VERSION_1, &info) !=
        ROCPROFILER_STATUS_SUCCESS) {
      continue;
    }
    (*by_name)[info.name] = id;
    (*instances)[id.handle] = info.dimensions_instances_count;
    g_names[id.handle] = info.name;
  }
  return true;
}

void sample_loop(int window_ms) {
  std::vector<rocprofiler_counter_record_t> records;
  size_t at = 0;
  while (!g_stop.load()) {
    Group& group = g_groups[at];
    at = (at + 1) % g_groups.size();
    if (!group.usable) {
      std::this_thread::sleep_for(std::chrono::milliseconds(5));
      continue;
    }
    g_current = group.config;
    const rocprofiler_status_t started = rocprofiler_start_context(g_context);
    if (started != ROCPROFILER_STATUS_SUCCESS) {  // (HSA not up yet: the application has not touched the GPU)
      std::this_thread::sleep_for(std::chrono::milliseconds(50));
      continue;
    }
    const double begin = now_s();
    std::this_thread::sleep_for(std::chrono::milliseconds(window_ms));
    records.assign(group.records, rocprofiler_counter_record_t{});
    size_t count = records.size();
    const rocprofiler_status_t sampled = rocprofiler_sample_device_counting_service(
        g_context, {}, ROCPROFILER_COUNTER_FLAG_NONE, records.data(), &count);
    const double length = now_s() - begin;
    rocprofiler_stop_context(g_context);
    if (sampled != ROCPROFILER_STATUS_SUCCESS) {
      static int complaints = 0;
      if (complaints++ < 3) ok(sampled, "sample");
      continue;
    }
    std::map<std::string, double> sums;
    for (size_t i = 0; i < count; ++i) {
      rocprofiler_counter_id_t id{};
      if (rocprofiler_query_record_counter_id(records[i].id, &id) != ROCPROFILER_STATUS_SUCCESS) continue;
      const auto name = g_names.find(id.handle);
      if (name != g_names.end()) sums[name->second] += records[i].counter_value;
    }
    for (const auto& [name, value] : sums) {
      std::fprintf(g_log, "%.6f %.6f %s %.0f\n", begin, length, name.c_str(), value);
    }
    std::fflush(g_log);
  }
  g_stopped.store(true);
}

int tool_init(rocprofiler_client_finalize_t finalize, void*) {
  g_finalize = finalize;
  if (!first_gpu(&g_agent)) {
    std::fprintf(stderr, "corun_counters: no GPU agent\n");
    return -1;
  }
  const char* log_name = std::getenv("AVR_COUNTER_LOG");
  g_log = std::fopen(log_name != nullptr ? log_name : "corun_counters.log", "w");
  if (g_log == nullptr) return -1;
  std::map<std::string, rocprofiler_counter_id_t> by_name;
  std::map<uint64_t, size_t> instances;
  if (!supported(&by_name, &instances)) return -1;
  const char* wanted = std::getenv("AVR_COUNTER_GROUPS");
  std::stringstream all(wanted != nullptr ? wanted
                                          : "SQ_WAVES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU");
  std::string group_text;
  while (std::getline(all, group_text, ';')) {
    Group group;
    std::vector<rocprofiler_counter_id_t> ids;
    std::stringstream names(group_text);
    std::string name;
    while (names >> name) {
      const auto found = by_name.find(name);
      if (found == by_name.end()) {
        std::fprintf(stderr, "corun_counters: no counter %s on this agent\n", name.c_str());
        continue;
      }
      group.names.push_back(name);
      ids.push_back(found->second);
      group.records += instances[found->second.handle];
    }
    if (ids.empty()) continue;
    group.usable = ok(rocprofiler_create_counter_config(g_agent, ids.data(), ids.size(), &group.config),
                      ("counter config of group '" + group_text + "'").c_str());
    g_groups.push_back(group);
  }
  if (g_groups.empty()) return -1;
  if (!ok(rocprofiler_create_context(&g_context), "context")) return -1;
  if (!ok(rocprofiler_create_buffer(g_context, 4096, 2048, ROCPROFILER_BUFFER_POLICY_LOSSLESS,
                                    [](rocprofiler_context_id_t, rocprofiler_buffer_id_t,
                                       rocprofiler_record_header_t**, size_t, void*, uint64_t) {},
                                    nullptr, &g_buffer),
          "buffer")) {
    return -1;
  }
  rocprofiler_callback_thread_t callbacks{};
  if (!ok(rocprofiler_create_callback_thread(&callbacks), "callback thread")) return -1;
  if (!ok(rocprofiler_assign_callback_thread(g_buffer, callbacks), "assign thread")) return -1;
  if (!ok(rocprofiler_configure_device_counting_service(
              g_context, g_buffer, g_agent,
              [](rocprofiler_context_id_t context, rocprofiler_agent_id_t,
                 rocprofiler_device_counting_agent_cb_t set_config, void*) {
                if (g_current.handle != 0) set_config(context, g_current);
              },
              nullptr),
          "device counting service")) {
    return -1;
  }
  const char* window = std::getenv("AVR_COUNTER_WINDOW_MS");
  const int window_ms = window != nullptr ? std::max(std::atoi(window), 10) : 250;
  g_thread = new std::thread(sample_loop, window_ms);
  return 0;
}

void tool_fini(void*) {
  g_client = nullptr;
  g_stop.store(true);
  if (g_thread != nullptr) {
    g_thread->join();
    delete g_thread;
    g_thread = nullptr;
  }
  if (g_log != nullptr) std::fclose(g_log);
  g_log = nullptr;
}

}  // namespace

extern "C" __attribute__((visibility("default"))) rocprofiler_tool_configure_result_t*
rocprofiler_configure(uint32_t, const char*, uint32_t, rocprofiler_client_id_t* id) {
  id->name = "avr_corun_counters";
  g_client = id;
  static rocprofiler_tool_configure_result_t result{sizeof(rocprofiler_tool_configure_result_t),
                                                    &tool_init, &tool_fini, nullptr};
  return &result;
}
