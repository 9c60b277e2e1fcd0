// config.hpp -- Config / Result / Saver of the `fora` command line.
// Mirrors the option names, defaults and JSON keys of the reference's config.h:86-308 and
// the argv scan of fora.cpp:98-160 (the observable contract), with a hand-written JSON
// writer in place of Boost.PropertyTree.
#pragma once
#include <cerrno>
#include <cstdint>
#include <cstdio>
#include <ctime>
#include <fstream>
#include <map>
#include <sstream>
#include <string>
#include <sys/stat.h>
#include <vector>

namespace forahost {

// action / algo names: config.h:33-45
static const char *const QUERY = "query";
static const char *const GEN_SS_QUERY = "generate-ss-query";
static const char *const TOPK = "topk";
static const char *const BUILD = "build";
static const char *const GEN_EXACT_TOPK = "gen-exact-topk"; // config.h:37
static const char *const BATCH_TOPK = "batch-topk";         // config.h:39
static const char *const CHECK_GRAPH = "check-graph"; // not in the reference: loader self-check
static const char *const CHECK_INDEX = "check-index"; // not in the reference: index file self-check (no GPU needed)
static const char *const FORA = "fora";

// timer slots: config.h:47-57
enum { FORA_QUERY = 3, FWD_LU = 5, RONDOM_WALK = 6, SORT_MAP = 8 };

struct Config { // config.h:86-160
    std::string graph_alias = "nethept"; // fora.cpp:63
    std::string graph_location;
    std::string action;
    std::string prefix = "./";
    std::string version = "vector";
    std::string exe_result_dir = "./"; // parent_folder, config.h:71
    bool multithread = false, with_rw_idx = false, opt = false, force_rebuild = false, balanced = false;
    double omega = 0, rmax = 0;
    unsigned int query_size = 1000; // config.h:113
    double pfail = 0, dbar = 0, epsilon = 0, delta = 0;
    unsigned int k = 500;          // config.h:122
    double ppr_decay_alpha = 0.77; // config.h:123
    double rw_cost_ratio = 8.0;    // config.h:125
    double rmax_scale = 1;         // config.h:127
    std::string algo;
    double alpha = 0.2;            // config.h:27,132
    std::string exact_pprs_folder;
    unsigned int hub_space_consum = 1;
    unsigned int max_iter_num = 100; // config.h:115
    // MI355X additions (not in the reference)
    uint64_t seed = 0x464F5241ull;
    int device = 0;
    int batch = 0;
    bool boost_idx = false;    // build --boost_idx: write the index as Boost binary archives (presumed 1.65 layout)
    bool oversubscribe = false; // --gpus N larger than the device count: several contexts per device (tests of the sharded driver)
    double balanced_start = 0; // --balanced_start S: first rmax of --balanced = S * rmax (0: the reference's 8)
    int gpus = 1;   // --gpus N: sources i mod N on GPU (device + i mod N), one host thread per GPU
    std::string get_graph_folder() const { return prefix + graph_alias + "/"; } // config.h:99-101
};

struct Result { // config.h:162-232
    int n = 0;
    long long m = 0;
    double avg_query_time = 0, total_mem_usage = 0, total_time_usage = 0;
    double num_randwalk = 0, num_rw_idx_use = 0, hit_idx_ratio = 0;
    double randwalk_time = 0, randwalk_time_ratio = 0, propagation_time = 0, propagation_time_ratio = 0;
    double topk_sort_time = 0;
    // config.h:188-196; the four error sums are never written by the reference either
    double topk_max_abs_err = 0, topk_avg_abs_err = 0, topk_max_relative_err = 0, topk_avg_relative_err = 0;
    double topk_precision = 0, topk_recall = 0;
    int real_topk_source_count = 0;
};

// "name=value" line on stdout like INFO(...) (mylib.h:557-559, mylib.cpp:11-21)
template <typename T> inline void info(const char *name, const T &v) {
    std::ostringstream os;
    os.precision(17);
    os << name << "=" << v;
    puts(os.str().c_str());
}

struct Timers { // Timer::timeUsed of mylib.h:594-657, seconds
    std::map<int, double> used;
    void add(int id, double s) { used[id] += s; }
    double get(int id) const { auto it = used.find(id); return it == used.end() ? 0.0 : it->second; }
    void show() const { // Timer::show
        for (auto &kv : used)
            if (kv.second > 0) printf("%d : %.6f s\n", kv.first, kv.second);
    }
};

inline std::string now_str() { // Saver::get_current_time_str, config.h:242-254
    time_t raw;
    time(&raw);
    char buf[80];
    strftime(buf, sizeof(buf), "%Y-%m-%d %H:%M:%S", localtime(&raw));
    return buf;
}

inline std::string jstr(const std::string &s) {
    std::string o = "\"";
    for (char ch : s) {
        if (ch == '"' || ch == '\\') { o += '\\'; o += ch; }
        else if (ch == '\n') o += "\\n";
        else o += ch;
    }
    return o + "\"";
}
template <typename T> inline std::string jnum(T v) {
    std::ostringstream os;
    os.precision(17);
    os << v;
    return "\"" + os.str() + "\""; // Boost ptree's write_json emits every value as a string
}

inline bool make_dirs(const std::string &path) {
    std::string cur;
    for (size_t i = 0; i < path.size(); i++) {
        cur += path[i];
        if (path[i] == '/' || i + 1 == path.size()) {
            if (cur != "/" && cur != "./" && mkdir(cur.c_str(), 0777) != 0 && errno != EEXIST) return false;
        }
    }
    return true;
}

// result file name: config.h:257-279
inline std::string result_path(Config &c) {
    if (c.exe_result_dir.empty() || c.exe_result_dir.back() != '/') c.exe_result_dir += "/";
    c.exe_result_dir += "execution/";
    make_dirs(c.exe_result_dir);
    std::string f = c.graph_alias + "." + c.action + "." + c.algo;
    f += std::string(".") + (c.with_rw_idx ? "with_idx" : "without_idx") + ".";
    f += "k-" + std::to_string(c.k) + ".";
    f += "rmax-" + std::to_string(c.rmax_scale);
    return c.exe_result_dir + f;
}

// Saver::save_json, config.h:288-307; keys of Config::get_data (:140-159) and Result::get_data (:198-230)
inline std::string save_json(Config &c, const Result &r, const Timers &t, const std::string &start_time,
                             const std::string &command_line) {
    const std::string path = result_path(c) + ".json";
    std::ofstream f(path);
    f << "{\n";
    f << "    \"start_time\": " << jstr(start_time) << ",\n";
    f << "    \"end_time\": " << jstr(now_str()) << ",\n";
    f << "    \"command_line\": " << jstr(command_line) << ",\n";
    f << "    \"config\": {\n";
    f << "        \"graph_alias\": " << jstr(c.graph_alias) << ",\n";
    f << "        \"action\": " << jstr(c.action) << ",\n";
    f << "        \"alpha\": " << jnum(c.alpha) << ",\n";
    f << "        \"pfail\": " << jnum(c.pfail) << ",\n";
    f << "        \"epsilon\": " << jnum(c.epsilon) << ",\n";
    f << "        \"delta\": " << jnum(c.delta) << ",\n";
    f << "        \"idx\": " << jstr(c.with_rw_idx ? "true" : "false") << ",\n";
    f << "        \"k\": " << jnum(c.k) << ",\n";
    f << "        \"rand-walk & push cost ratio\": " << jnum(c.rw_cost_ratio) << ",\n";
    f << "        \"query-size\": " << jnum(c.query_size) << ",\n";
    f << "        \"algo\": " << jstr(c.algo) << ",\n";
    f << "        \"rmax\": " << jnum(c.rmax) << ",\n";
    f << "        \"rmax-scale\": " << jnum(c.rmax_scale) << ",\n";
    f << "        \"omega\": " << jnum(c.omega) << ",\n";
    f << "        \"result-dir\": " << jstr(c.exe_result_dir) << "\n";
    f << "    },\n";
    f << "    \"result\": {\n";
    f << "        \"n\": " << jnum(r.n) << ",\n";
    f << "        \"m\": " << jnum(r.m) << ",\n";
    f << "        \"avg query time(s/q)\": " << jnum(r.avg_query_time) << ",\n";
    f << "        \"total memory usage(MB)\": " << jnum(r.total_mem_usage) << ",\n";
    f << "        \"total time usage(s)\": " << jnum(r.total_time_usage) << ",\n";
    f << "        \"total time on rand-walks(s)\": " << jnum(r.randwalk_time) << ",\n";
    f << "        \"total time on propagation(s)\": " << jnum(r.propagation_time) << ",\n";
    f << "        \"total time on sorting top-k ppr(s)\": " << jnum(r.topk_sort_time) << ",\n";
    f << "        \"total time ratio on rand-walks(%)\": " << jnum(r.randwalk_time_ratio) << ",\n";
    f << "        \"total time ratio on propagation(%)\": " << jnum(r.propagation_time_ratio) << ",\n";
    f << "        \"total number of rand-walks\": " << jnum(r.num_randwalk) << ",\n";
    f << "        \"total number of rand-walk idx used\": " << jnum(r.num_rw_idx_use) << ",\n";
    f << "        \"total usage ratio of rand-walk idx\": " << jnum(r.hit_idx_ratio) << ",\n";
    const double cnt = (double)r.real_topk_source_count; // config.h:223-228 divide by it unguarded: nan when no source had ground truth
    f << "        \"topk max absolute error\": " << jnum(r.topk_max_abs_err / cnt) << ",\n";
    f << "        \"topk avg absolute error\": " << jnum(r.topk_avg_abs_err / cnt) << ",\n";
    f << "        \"topk max relative error\": " << jnum(r.topk_max_relative_err / cnt) << ",\n";
    f << "        \"topk avg relative error\": " << jnum(r.topk_avg_relative_err / cnt) << ",\n";
    f << "        \"topk precision\": " << jnum(r.topk_precision / cnt) << ",\n";
    f << "        \"topk recall\": " << jnum(r.topk_recall / cnt) << "\n";
    f << "    },\n";
    f << "    \"timer\": {\n";
    bool first = true;
    for (auto &kv : t.used) {
        if (kv.second <= 0) continue;
        if (!first) f << ",\n";
        f << "        \"" << kv.first << "\": " << jnum(kv.second);
        first = false;
    }
    f << "\n    }\n}\n";
    return path;
}

} // namespace forahost
