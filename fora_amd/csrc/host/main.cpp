// fora -- command line of the MI355X-native FORA engine: `fora query|topk|build ...`.
// Drop-in for the reference's main (fora.cpp:56-292) on the FORA path: same actions, flag
// names, input files and result JSON; the per-source loops of query() (query.h:1460-1481) and
// topk() (query.h:1397-1401) become batched calls through the C ABI of libfora_hip.so.
#include "config.hpp"
#include "graph.hpp"
#include "fora_hip.h"

#include <chrono>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <iostream>
#include <map>
#include <unordered_map>
#include <unordered_set>
#include <sys/resource.h>
#include <sys/stat.h>
#include <thread>

using namespace forahost;
using std::cerr;
using std::cout;
using std::endl;
using std::string;

static Config config;
static Result result;
static Timers timers;

static double now_s() {
    return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}
static double proc_memory_mb() { // get_proc_memory()/1000.0, mylib.h:695-699
    struct rusage u;
    getrusage(RUSAGE_SELF, &u);
    return u.ru_maxrss / 1000.0;
}
static void split_line() { cout << "-----------------------------------------------" << endl; }

static const char *HELP =
    "fora query --algo <algo> [options]\n"
    "fora topk  --algo <algo> [options]\n"
    "fora build [options]\n"
    "fora gen-exact-topk [options]\n"
    "fora batch-topk --algo <algo> [options]\n"
    "fora generate-ss-query [options]\n"
    "fora\n"
    "\n"
    "algo: \n"
    "  fora\n"
    "options: \n"
    "  --prefix <prefix>\n"
    "  --epsilon <epsilon>\n"
    "  --dataset <dataset>\n"
    "  --query_size <queries count>\n"
    "  --k <top k>\n"
    "  --with_idx\n"
    "  --opt\n"
    "  --balanced\n"
    "  --result_dir <directory to place results>\n"
    "  --rmax_scale <scale of rmax>\n"
    "  --seed <walk RNG seed>   (MI355X build: Philox, reproducible)\n"
    "  --device <gpu ordinal>   --batch <queries in flight>\n"
    "  --gpus <N>               (sources i mod N on N GPUs of this node, one host thread per GPU)\n";

#define FAIL(ctx, what)                                                                        \
    do {                                                                                       \
        cerr << what << ": " << (ctx ? fora_hip_last_error(ctx) : "no context") << endl;      \
        return 1;                                                                              \
    } while (0)

// display_time_usage (algo.h:368-402) + set_result (algo.h:404-440)
static void finish(const Graph &graph, int used_counter, unsigned query_size, double n_walks, double n_hit) {
    const double tot = timers.get(used_counter);
    cout << "Total cost (s): " << tot << endl;
    cout << timers.get(RONDOM_WALK) * 100.0 / tot << "%" << " for random walk cost" << endl;
    cout << timers.get(FWD_LU) * 100.0 / tot << "%" << " for forward push cost" << endl;
    split_line();
    if (config.with_rw_idx) cout << "Average rand-walk idx hit ratio: " << n_hit * 100.0 / n_walks << "%" << endl;
    if (config.action == TOPK) { // algo.h:394-398; the reference asserts when no source had ground truth
        if (result.real_topk_source_count > 0) {
            cout << "Average top-K Precision: " << result.topk_precision / result.real_topk_source_count << endl;
            cout << "Average top-K Recall: " << result.topk_recall / result.real_topk_source_count << endl;
        } else cout << "no exact top-k file (fora gen-exact-topk): precision / recall not evaluated" << endl;
    }
    cout << "Average query time (s):" << tot / query_size << endl;
    cout << "Memory usage (MB):" << proc_memory_mb() << endl << endl;
    config.query_size = query_size;
    result.m = graph.m; result.n = graph.n;
    result.avg_query_time = tot / query_size;
    result.total_mem_usage = proc_memory_mb();
    result.total_time_usage = tot;
    result.num_randwalk = n_walks;
    if (config.with_rw_idx) { result.num_rw_idx_use = n_hit; result.hit_idx_ratio = n_hit / n_walks; }
    result.randwalk_time = timers.get(RONDOM_WALK);
    result.randwalk_time_ratio = timers.get(RONDOM_WALK) * 100 / tot;
    result.propagation_time = timers.get(FWD_LU);
    result.propagation_time_ratio = timers.get(FWD_LU) * 100 / tot;
    if (config.action == TOPK) result.topk_sort_time = timers.get(SORT_MAP);
}

// ---- ground truth for top-k: exact_topk_pprs (algo.h:45), <exact_pprs_folder>/<dataset>.topk.pprs (build.h:121-125).
// The reference stores the map as a Boost text archive; this build keeps the file name and writes one line per
// source: "<source> <count> <id>:<score> ..." under a one-line header.
using ExactTopk = std::map<int32_t, std::vector<std::pair<int32_t, double>>>;
static ExactTopk exact_topk_pprs;

static string exact_topk_file() {
    if (config.exact_pprs_folder.empty() || config.exact_pprs_folder.back() != '/') config.exact_pprs_folder += "/";
    return config.exact_pprs_folder + config.graph_alias + ".topk.pprs";
}
static bool file_exists(const string &f) {
    if (FILE *t = fopen(f.c_str(), "r")) { fclose(t); return true; }
    return false;
}
static bool save_exact_topk(const string &f) { // save_exact_topk_ppr, build.h:127-132
    FILE *fo = fopen(f.c_str(), "w");
    if (!fo) return false;
    fprintf(fo, "fora-exact-topk 1 %zu\n", exact_topk_pprs.size());
    for (auto &kv : exact_topk_pprs) {
        fprintf(fo, "%d %zu", kv.first, kv.second.size());
        for (auto &p : kv.second) fprintf(fo, " %d:%.17g", p.first, p.second);
        fprintf(fo, "\n");
    }
    return fclose(fo) == 0;
}
static void load_exact_topk() { // load_exact_topk_ppr, build.h:134-145
    const string f = exact_topk_file();
    if (!file_exists(f)) { info("No exact topk ppr file", f); return; }
    FILE *fi = fopen(f.c_str(), "r");
    char tag[32] = {0};
    int ver = 0;
    size_t count = 0;
    if (!fi || fscanf(fi, "%31s %d %zu", tag, &ver, &count) != 3 || string(tag) != "fora-exact-topk" || ver != 1) {
        cerr << f << " is not a fora-exact-topk v1 file (Boost text archives of the reference are not readable here)" << endl;
        if (fi) fclose(fi);
        return;
    }
    for (size_t i = 0; i < count; i++) {
        int src;
        size_t len;
        if (fscanf(fi, "%d %zu", &src, &len) != 2) break;
        auto &v = exact_topk_pprs[src];
        v.resize(len);
        for (size_t j = 0; j < len; j++)
            if (fscanf(fi, "%d:%lf", &v[j].first, &v[j].second) != 2) { v.resize(j); break; }
    }
    fclose(fi);
    info("exact_topk_pprs.size()", exact_topk_pprs.size());
}

// compute_precision, algo.h:524-572 (both ratios are over the size of the exact set, as in the reference)
static void compute_precision(int32_t v, const int32_t *ids, const double *scores, unsigned k) {
    auto it = exact_topk_pprs.find(v);
    if (exact_topk_pprs.empty() || it == exact_topk_pprs.end()) return;
    std::unordered_set<int32_t> topk_set, exact_set;
    for (unsigned i = 0; i < k; i++)
        if (scores[i] > 0) topk_set.insert(ids[i]);
    double recall = 0, precision = 0;
    const size_t size_e = std::min<size_t>(k, it->second.size());
    for (size_t i = 0; i < size_e; i++) {
        const auto &p = it->second[i];
        if (p.second > 0) {
            exact_set.insert(p.first);
            if (topk_set.count(p.first)) recall++;
        }
    }
    for (int32_t id : topk_set)
        if (exact_set.count(id)) precision++;
    if (exact_set.empty() || topk_set.empty()) return; // the reference asserts here (algo.h:559-560)
    recall /= (double)exact_set.size();
    precision /= (double)exact_set.size();
    cout << "exact_map.size()=" << exact_set.size() << " recall=" << recall << " precision=" << precision << endl;
    result.topk_recall += recall;
    result.topk_precision += precision;
    result.real_topk_source_count++;
}

static int open_engine(const Graph &graph, fora_ctx **ctx, int device = -1) {
    int rc = fora_hip_create(device < 0 ? config.device : device, ctx);
    if (rc) { cerr << "no usable MI355X (gfx950) device: fora_hip_create rc=" << rc << endl; return 1; }
    if (fora_hip_set_graph(*ctx, graph.n, graph.m, graph.row_ptr.data(), graph.col.data())) FAIL(*ctx, "set_graph");
    if (fora_hip_set_params(*ctx, config.alpha, config.epsilon, config.rmax_scale, config.opt, config.seed)) FAIL(*ctx, "set_params");
    fora_hip_get_params(*ctx, &config.rmax, &config.omega); // fora_setting, algo.h:455-463
    if (config.batch) fora_hip_set_batch(*ctx, config.batch);
    return 0;
}

// ---- multi-GPU: one host thread and one context per GPU, source i of the query list on shard i mod G
// (the per-source loops of query.h:1471-1476 / 1397-1401 carry no state from one source to the next)
struct IndexData { std::vector<int32_t> rw; std::vector<uint64_t> off, cnt; };
struct Shard {
    std::vector<int32_t> sources;
    std::vector<unsigned> pos; // position of each source in the query list
    fora_timing tm{};
    double seconds = 0;
    int rc = 0;
    string err;
};

template <class Work> // Work(ctx, shard) -> rc, runs the shard's batch call and scatters its outputs
static int run_sharded(const Graph &graph, const std::vector<int32_t> &queries, unsigned query_size,
                       const IndexData *index, std::vector<Shard> &shards, Work work) {
    int ndev = fora_hip_device_count();
    if (ndev <= 0) { cerr << "no usable MI355X (gfx950) device" << endl; return 1; }
    if (config.gpus > ndev && !config.oversubscribe) { // two contexts on one GPU would each size their batch from 40 % of its HBM
        cerr << "--gpus " << config.gpus << " but only " << ndev << " device(s) visible; using " << ndev << endl;
        config.gpus = ndev;
    }
    const int G = std::max(1, config.gpus);
    shards.assign((size_t)G, Shard());
    for (unsigned i = 0; i < query_size; i++) {
        shards[i % G].sources.push_back(queries[i]);
        shards[i % G].pos.push_back(i);
    }
    std::vector<std::thread> threads;
    for (int g = 0; g < G; g++)
        threads.emplace_back([&, g]() {
            Shard &s = shards[(size_t)g];
            fora_ctx *ctx = nullptr;
            if (fora_hip_create((config.device + g) % ndev, &ctx)) { s.rc = 1; s.err = "fora_hip_create failed"; return; }
            auto bail = [&](const char *what) { s.rc = 1; s.err = string(what) + ": " + fora_hip_last_error(ctx); fora_hip_destroy(ctx); };
            if (fora_hip_set_graph(ctx, graph.n, graph.m, graph.row_ptr.data(), graph.col.data())) return bail("set_graph");
            if (fora_hip_set_params(ctx, config.alpha, config.epsilon, config.rmax_scale, config.opt, config.seed)) return bail("set_params");
            if (config.batch) fora_hip_set_batch(ctx, config.batch);
            if (config.balanced) fora_hip_set_balanced(ctx, 1, config.balanced_start, 0, 0, 0, 0); // query.h:848-884, MI355X cost model
            if (index && fora_hip_set_index(ctx, index->rw.data(), index->rw.size(), index->off.data(), index->cnt.data()))
                return bail("set_index");
            fora_hip_reset_timing(ctx);
            const double t0 = now_s();
            if (work(ctx, s)) return bail("run");
            s.seconds = now_s() - t0;
            fora_hip_get_timing(ctx, &s.tm);
            fora_hip_destroy(ctx);
        });
    for (auto &t : threads) t.join();
    for (auto &s : shards)
        if (s.rc) { cerr << s.err << endl; return 1; }
    return 0;
}

static void add_shard_timers(const std::vector<Shard> &shards, int total_slot) {
    double wall = 0, push = 0, walk = 0, other = 0;
    for (auto &s : shards) { // shards run concurrently: the slowest one is the elapsed time
        wall = std::max(wall, s.seconds);
        // every push kernel of the engine: direct (pop + expand), bucketed (bin = "expand" + accumulate), team, tail
        const double push_ms = s.tm.push_pop_ms + s.tm.push_expand_ms + s.tm.push_accum_ms + s.tm.push_tail_ms + s.tm.push_team_ms;
        push = std::max(push, push_ms * 1e-3);
        if (getenv("FORA_CLI_TIMING")) // tests: the engine's own sums beside the reference's timer slots
            fprintf(stderr, "engine_timing push_ms=%.6f walk_ms=%.6f other_ms=%.6f batch_ms=%.6f wall_s=%.6f\n", push_ms,
                    s.tm.walk_alloc_ms + s.tm.walk_ms + s.tm.walk_accum_ms, s.tm.other_ms, s.tm.batch_ms, s.seconds);
        walk = std::max(walk, (s.tm.walk_alloc_ms + s.tm.walk_ms + s.tm.walk_accum_ms) * 1e-3);
        other = std::max(other, s.tm.other_ms * 1e-3);
    }
    timers.add(total_slot, wall);
    timers.add(FWD_LU, push);
    timers.add(RONDOM_WALK, walk);
    if (config.action == TOPK) timers.add(SORT_MAP, other);
}

static int do_query(Graph &graph) { // query(), query.h:1415-1515 FORA branch
    info("config.algo", config.algo);
    std::vector<int32_t> queries;
    if (!graph.load_ss_query(queries)) { cerr << graph.error << endl; exit(0); } // algo.h:513-516
    unsigned query_size = std::min<unsigned>((unsigned)queries.size(), config.query_size);
    info("query_size", query_size);
    if (!(config.rmax_scale >= 0)) { cerr << "rmax_scale must be >= 0" << endl; return 1; } // query.h:1424
    { // fora_setting (algo.h:455-463) for the log lines / result JSON; the engine recomputes the same values
        const double delta = 1.0 / graph.n, pfail = 1.0 / graph.n;
        config.rmax = config.epsilon * sqrt(delta / 3 / graph.m / log(2 / pfail));
        config.rmax *= config.opt ? config.rmax_scale / (1 - config.alpha) : config.rmax_scale;
        config.omega = (2 + config.epsilon) * log(2 / pfail) / delta / config.epsilon / config.epsilon;
    }
    info("config.rmax", config.rmax);
    info("config.omega", config.omega);
    IndexData index;
    if (config.with_rw_idx) { // deserialize_idx, build.h:194-207
        string err;
        if (!IndexFile::read(config.graph_location, config.rmax_scale, config.opt, graph.n, index.rw, index.off, index.cnt, err)) {
            cerr << err << endl;
            return 1;
        }
    }
    std::vector<fora_query_stats> st(query_size);
    std::vector<Shard> shards;
    if (run_sharded(graph, queries, query_size, config.with_rw_idx ? &index : nullptr, shards, [&](fora_ctx *ctx, Shard &s) {
            std::vector<fora_query_stats> local(s.sources.size());
            if (fora_hip_query_batch(ctx, s.sources.data(), (int)s.sources.size(), config.with_rw_idx, nullptr, local.data())) return 1;
            for (size_t i = 0; i < local.size(); i++) st[s.pos[i]] = local[i];
            return 0;
        }))
        return 1;
    double n_walks = 0, n_hit = 0, total_rsum = 0;
    for (unsigned i = 0; i < query_size; i++) {
        cout << i + 1 << ". source node:" << queries[i] << endl; // query.h:1473
        n_walks += (double)st[i].n_walks;
        n_hit += (double)st[i].n_idx_hit;
        total_rsum += st[i].rsum * (1 - config.alpha); // query.h:896,899
    }
    split_line();
    info("avg_rsum*config.omega", total_rsum / query_size * config.omega);
    add_shard_timers(shards, FORA_QUERY);
    finish(graph, FORA_QUERY, query_size, n_walks, n_hit);
    return 0;
}

// one pass of get_topk (query.h:1139-1156) over the query list for the current config.k; lists come back in query order
static int run_topk_pass(const Graph &graph, const std::vector<int32_t> &queries, unsigned query_size, const IndexData *index,
                         std::vector<int32_t> &ids, std::vector<double> &scores, std::vector<int32_t> &rounds,
                         std::vector<Shard> &shards) {
    const size_t k = config.k;
    ids.assign((size_t)query_size * k, 0);
    scores.assign((size_t)query_size * k, 0.0);
    rounds.assign(query_size, 0);
    return run_sharded(graph, queries, query_size, index, shards, [&](fora_ctx *ctx, Shard &s) {
        const size_t nl = s.sources.size();
        std::vector<int32_t> lid(nl * k), lr(nl);
        std::vector<double> lsc(nl * k);
        // get_topk, query.h:1150-1153: --opt -> fora_query_topk_new, else fora_query_topk_with_bound
        const int rc = config.opt ? fora_hip_topk_batch(ctx, s.sources.data(), (int)nl, (int)k, config.epsilon, config.rmax_scale,
                                                        config.with_rw_idx, lid.data(), lsc.data(), lr.data())
                                  : fora_hip_topk_bound_batch(ctx, s.sources.data(), (int)nl, (int)k, config.epsilon,
                                                              config.rmax_scale, config.ppr_decay_alpha, config.with_rw_idx,
                                                              lid.data(), lsc.data(), lr.data());
        if (rc) return 1;
        for (size_t i = 0; i < nl; i++) { // the "gather": top-k lists back in query order
            std::copy(lid.begin() + (long)(i * k), lid.begin() + (long)((i + 1) * k), ids.begin() + (long)(s.pos[i] * k));
            std::copy(lsc.begin() + (long)(i * k), lsc.begin() + (long)((i + 1) * k), scores.begin() + (long)(s.pos[i] * k));
            rounds[s.pos[i]] = lr[i];
        }
        return 0;
    });
}

static int do_topk(Graph &graph) { // topk(), query.h:1309-1413 FORA branch
    std::vector<int32_t> queries;
    if (!graph.load_ss_query(queries)) { cerr << graph.error << endl; exit(0); }
    info("queries.size()", queries.size());
    unsigned query_size = std::min<unsigned>((unsigned)queries.size(), config.query_size);
    if (!(config.k < (unsigned)graph.n - 1) || !(config.k > 1)) { cerr << "k out of range" << endl; return 1; } // :1317-1318
    info("config.k", config.k);
    split_line();
    IndexData index;
    if (config.with_rw_idx) {
        string err;
        if (!IndexFile::read(config.graph_location, config.rmax_scale, config.opt, graph.n, index.rw, index.off, index.cnt, err)) {
            cerr << err << endl;
            return 1;
        }
    }
    load_exact_topk(); // query.h:1323
    std::vector<int32_t> ids, rounds;
    std::vector<double> scores;
    std::vector<Shard> shards;
    if (run_topk_pass(graph, queries, query_size, config.with_rw_idx ? &index : nullptr, ids, scores, rounds, shards)) return 1;
    double tot_walks = 0, tot_hits = 0;
    for (auto &s : shards) { tot_walks += (double)s.tm.walks; tot_hits += (double)s.tm.idx_hits; }
    long num_iter_topk = 0;
    if (config.exe_result_dir.empty() || config.exe_result_dir.back() != '/') config.exe_result_dir += "/";
    make_dirs(config.exe_result_dir);
    const string out = config.exe_result_dir + config.graph_alias + ".topk.k-" + std::to_string(config.k) + ".txt";
    FILE *fo = fopen(out.c_str(), "w");
    for (unsigned i = 0; i < query_size; i++) {
        cout << i + 1 << ". source node:" << queries[i] << endl; // query.h:1398
        num_iter_topk += rounds[i];
        compute_precision(queries[i], &ids[(size_t)i * config.k], &scores[(size_t)i * config.k], config.k); // query.h:1180
        if (fo) {
            fprintf(fo, "%d", queries[i]);
            for (unsigned j = 0; j < config.k; j++) fprintf(fo, " %d:%.17g", ids[(size_t)i * config.k + j], scores[(size_t)i * config.k + j]);
            fprintf(fo, "\n");
        }
    }
    if (fo) fclose(fo);
    split_line();
    cout << "average iter times:" << num_iter_topk / query_size << endl; // query.h:1403
    add_shard_timers(shards, FORA_QUERY);
    timers.add(0, timers.get(FORA_QUERY));
    finish(graph, FORA_QUERY, query_size, tot_walks, tot_hits);
    cout << "top-k lists written to " << out << endl;
    return 0;
}

static int do_batch_topk(Graph &graph) { // batch_topk(), query.h:1517-1640, FORA branch (:1612-1636): the algorithm runs again per k
    std::vector<int32_t> queries;
    if (!graph.load_ss_query(queries)) { cerr << graph.error << endl; exit(0); }
    info("queries.size()", queries.size());
    unsigned query_size = std::min<unsigned>((unsigned)queries.size(), config.query_size);
    if (!(config.k < (unsigned)graph.n - 1) || !(config.k > 1)) { cerr << "k out of range" << endl; return 1; } // :1524-1525
    info("config.k", config.k);
    split_line();
    load_exact_topk(); // :1529
    IndexData index;
    if (config.with_rw_idx) {
        string err;
        if (!IndexFile::read(config.graph_location, config.rmax_scale, config.opt, graph.n, index.rw, index.off, index.cnt, err)) {
            cerr << err << endl;
            return 1;
        }
    }
    std::vector<unsigned> ks; // :1585-1591
    const unsigned step = config.k / 5;
    if (step > 0) for (unsigned i = 1; i < 5; i++) if (i * step >= 2) ks.push_back(i * step); // get_topk needs k > 1
    ks.push_back(config.k);
    struct Pred { double precision = 0, recall = 0; int count = 0; };
    std::map<unsigned, Pred> pred;
    for (unsigned k : ks) {
        if (k < 2) continue; // get_topk needs k > 1
        config.k = k;
        info("========================================", "");
        info("k is set to be ", config.k);
        result.topk_recall = 0; result.topk_precision = 0; result.real_topk_source_count = 0;
        std::vector<int32_t> ids, rounds;
        std::vector<double> scores;
        std::vector<Shard> shards;
        if (run_topk_pass(graph, queries, query_size, config.with_rw_idx ? &index : nullptr, ids, scores, rounds, shards)) return 1;
        double wall = 0;
        for (auto &s : shards) wall = std::max(wall, s.seconds);
        for (unsigned i = 0; i < query_size; i++) {
            cout << i + 1 << ". source node:" << queries[i] << endl;
            compute_precision(queries[i], &ids[(size_t)i * k], &scores[(size_t)i * k], k);
        }
        pred[k].precision = result.topk_precision; pred[k].recall = result.topk_recall; pred[k].count = result.real_topk_source_count;
        cout << "k=" << k << " precision=" << result.topk_precision / result.real_topk_source_count
             << " recall=" << result.topk_recall / result.real_topk_source_count << endl;   // :1628-1629
        cout << "Average query time (s):" << wall / query_size << endl;                   // :1630
    }
    split_line(); // display_precision_for_dif_k, algo.h:676-692
    cout << config.algo << endl;
    for (unsigned k : ks) cout << k << "\t";
    cout << endl << "Precision:" << endl;
    for (unsigned k : ks) cout << (pred[k].count ? pred[k].precision / pred[k].count : 0.0) << "\t";
    cout << endl << "Recall:" << endl;
    for (unsigned k : ks) cout << (pred[k].count ? pred[k].recall / pred[k].count : 0.0) << "\t";
    cout << endl;
    return 0;
}

static int do_gen_exact_topk(Graph &graph) { // gen_exact_topk(), query.h:1240-1307
    std::vector<int32_t> queries;
    if (!graph.load_ss_query(queries)) { cerr << graph.error << endl; exit(0); }
    info("queries.size()", queries.size());
    unsigned query_size = std::min<unsigned>((unsigned)queries.size(), config.query_size);
    info("query_size", query_size);
    const string f = exact_topk_file();
    if (file_exists(f)) { puts("exact top k exists"); return 0; } // query.h:1251-1254
    if (!(config.k < (unsigned)graph.n - 1) || !(config.k > 1)) { cerr << "k out of range" << endl; return 1; } // :1255-1256
    if (config.k > 1024) { cerr << "k > 1024 not supported" << endl; return 1; }
    info("config.k", config.k);
    split_line();
    puts("power itrating...");
    if (!(config.epsilon > 0)) config.epsilon = 0.5; // not used by the iteration; the engine wants a valid parameter set
    const size_t k = config.k;
    std::vector<int32_t> ids((size_t)query_size * k);
    std::vector<double> scores((size_t)query_size * k);
    std::vector<Shard> shards;
    // multi_power_iter (query.h:1226-1238) threads over CPU cores; here sources i mod G over the GPUs
    if (run_sharded(graph, queries, query_size, nullptr, shards, [&](fora_ctx *ctx, Shard &s) {
            const size_t nl = s.sources.size();
            std::vector<int32_t> lid(nl * k);
            std::vector<double> lsc(nl * k);
            if (fora_hip_power_iteration_batch(ctx, s.sources.data(), (int)nl, (int)config.max_iter_num, nullptr, nullptr, (int)k,
                                               lid.data(), lsc.data()))
                return 1;
            for (size_t i = 0; i < nl; i++) {
                std::copy(lid.begin() + (long)(i * k), lid.begin() + (long)((i + 1) * k), ids.begin() + (long)(s.pos[i] * k));
                std::copy(lsc.begin() + (long)(i * k), lsc.begin() + (long)((i + 1) * k), scores.begin() + (long)(s.pos[i] * k));
            }
            return 0;
        }))
        return 1;
    double wall = 0;
    for (auto &s : shards) wall = std::max(wall, s.seconds);
    timers.add(10, wall); // PI_QUERY, config.h:57
    cout << "average generation time (s): " << wall / query_size << endl; // query.h:1294
    puts("combine results...");
    for (unsigned i = 0; i < query_size; i++) {
        auto &v = exact_topk_pprs[queries[i]];
        v.resize(k);
        for (size_t j = 0; j < k; j++) v[j] = {ids[(size_t)i * k + j], scores[(size_t)i * k + j]};
    }
    make_dirs(config.exact_pprs_folder);
    if (!save_exact_topk(f)) { cerr << "cannot write " << f << endl; return 1; }
    cout << "exact top-k lists written to " << f << endl;
    return 0;
}

static int do_build(Graph &graph) { // build(), build.h:302-366
    fora_ctx *ctx = nullptr;
    if (open_engine(graph, &ctx)) return 1;
    uint64_t total = 0;
    if (fora_hip_index_sizes(ctx, &total, nullptr, nullptr)) FAIL(ctx, "index_sizes");
    info("tuned_index_size", total);
    puts("rand-walking...");
    info("config.rmax", config.rmax);
    info("config.omega", config.omega);
    info("config.rmax*config.omega", config.rmax * config.omega);
    double t0 = now_s();
    if (fora_hip_build_index(ctx)) FAIL(ctx, "build_index");
    timers.add(1, now_s() - t0);
    puts("materializing...");
    t0 = now_s();
    std::vector<int32_t> rw(std::max<uint64_t>(1, total));
    std::vector<uint64_t> off((size_t)graph.n), cnt((size_t)graph.n);
    if (fora_hip_get_index(ctx, rw.data(), total, off.data(), cnt.data())) FAIL(ctx, "get_index");
    rw.resize(total);
    info("rw_idx.size()", rw.size());
    string err;
    if (config.boost_idx)
        cerr << "warning: --boost_idx writes the layout Boost's binary_oarchive is documented to produce; no reference-written "
                "file was available to check it against (experimental, see INTEGRATION.md 7)" << endl;
    const bool wrote = config.boost_idx ? IndexFile::write_boost(config.graph_location, config.rmax_scale, config.opt, graph.n, rw, off, cnt, err)
                                        : IndexFile::write(config.graph_location, config.rmax_scale, config.opt, graph.n, rw, off, cnt, err);
    if (!wrote) {
        cerr << err << endl;
        fora_hip_destroy(ctx);
        return 1;
    }
    timers.add(2, now_s() - t0);
    cout << "Memory usage (MB):" << proc_memory_mb() << endl << endl;
    fora_hip_destroy(ctx);
    return 0;
}

int main(int argc, char *argv[]) {
    const string start_time = now_str();
    for (int i = 0; i < argc; i++)
        if (string(argv[i]) == "--help") { cout << HELP << endl; return 0; } // fora.cpp:93-96
    if (argc < 2) { cerr << "sub command not regoznized" << endl; return 1; }
    config.action = argv[1]; // fora.cpp:98
    cout << "action: " << config.action << endl;
    for (int i = 0; i < argc; i++) { // fora.cpp:111-160
        const string arg = argv[i];
        auto next = [&](const char *what) -> const char * {
            if (i + 1 >= argc) { cerr << "missing value for " << what << endl; exit(1); }
            return argv[i + 1];
        };
        if (arg == "--algo") config.algo = next("--algo");
        else if (arg == "--epsilon") { config.epsilon = atof(next("--epsilon")); info("config.epsilon", config.epsilon); }
        else if (arg == "--multithread") config.multithread = true;
        else if (arg == "--result_dir") config.exe_result_dir = next("--result_dir");
        else if (arg == "--exact_ppr_path") config.exact_pprs_folder = next("--exact_ppr_path");
        else if (arg == "--with_idx") config.with_rw_idx = true;
        else if (arg == "--rmax_scale") config.rmax_scale = atof(next("--rmax_scale"));
        else if (arg == "--force-rebuild") config.force_rebuild = true;
        else if (arg == "--query_size") config.query_size = (unsigned)atoi(next("--query_size"));
        else if (arg == "--hub_space") config.hub_space_consum = (unsigned)atoi(next("--hub_space"));
        else if (arg == "--version") config.version = next("--version");
        else if (arg == "--k") config.k = (unsigned)atoi(next("--k"));
        else if (arg == "--rw_ratio") config.rw_cost_ratio = atof(next("--rw_ratio"));
        else if (arg == "--prefix") config.prefix = next("--prefix");
        else if (arg == "--dataset") config.graph_alias = next("--dataset");
        else if (arg == "--opt") config.opt = true;
        else if (arg == "--balanced") config.balanced = true;
        else if (arg == "--boost_idx") config.boost_idx = true;
        else if (arg == "--oversubscribe") config.oversubscribe = true; // tests: --gpus N contexts on fewer devices
        else if (arg == "--balanced_start") config.balanced_start = atof(next("--balanced_start"));
        else if (arg == "--seed") config.seed = strtoull(next("--seed"), nullptr, 0);
        else if (arg == "--device") config.device = atoi(next("--device"));
        else if (arg == "--batch") config.batch = atoi(next("--batch"));
        else if (arg == "--gpus") config.gpus = std::max(1, atoi(next("--gpus")));
        else if (arg.substr(0, 2) == "--") { cerr << "command not recognize " << arg << endl; return 1; } // fora.cpp:156-159
    }
    info("config.version", config.version);
    info("config.action", config.action);

    const string act = config.action;
    if (act != QUERY && act != TOPK && act != BUILD && act != GEN_SS_QUERY && act != CHECK_GRAPH && act != GEN_EXACT_TOPK &&
        act != CHECK_INDEX && act != BATCH_TOPK) {
        cerr << "sub command not regoznized" << endl; // fora.cpp:278-281
        return 1;
    }
    if ((act == QUERY || act == TOPK || act == BATCH_TOPK) && config.algo != FORA) { // fora.cpp:169-175, :226-231
        info("Wrong algo param: ", config.algo);
        cerr << "only --algo fora is part of this build" << endl;
        return 1;
    }
    if ((act == QUERY || act == TOPK || act == BUILD || act == BATCH_TOPK) && !(config.epsilon > 0)) { cerr << "--epsilon must be > 0" << endl; return 1; }
    config.graph_location = config.get_graph_folder();
    Graph graph;
    graph.data_folder = config.graph_location;
    const bool ok = (act == GEN_SS_QUERY || act == CHECK_INDEX) ? graph.init_nm() : graph.init_graph(); // graph.h:40-43
    if (!ok) { cerr << graph.error << endl; return 1; }
    cout << "init graph n: " << graph.n << " m: " << graph.m << endl;
    config.delta = 1.0 / graph.n; // init_parameter, graph.h:173-183
    config.pfail = 1.0 / graph.n;
    config.dbar = double(graph.m) / double(graph.n);
    info("graph.n", graph.n);
    info("graph.m", graph.m);

    { // fora.cpp:211-212, 254-255
        struct stat sb;
        if (config.exact_pprs_folder.empty() || stat(config.exact_pprs_folder.c_str(), &sb) != 0) config.exact_pprs_folder = config.graph_location;
    }
    int rc = 0;
    if (act == CHECK_INDEX) { // reads randwalks.idx/.info (this build's container or a Boost archive of the reference)
        IndexData ix;
        string err;
        if (!IndexFile::read(config.graph_location, config.rmax_scale, config.opt, graph.n, ix.rw, ix.off, ix.cnt, err)) {
            cerr << err << endl;
            return 1;
        }
        uint64_t h = 1469598103934665603ull;
        for (int32_t v : ix.rw) { h ^= (uint32_t)v; h *= 1099511628211ull; }
        for (size_t v = 0; v < ix.cnt.size(); v++) { h ^= ix.off[v]; h *= 1099511628211ull; h ^= ix.cnt[v]; h *= 1099511628211ull; }
        cout << "index walks: " << ix.rw.size() << " nodes: " << ix.cnt.size() << " fnv1a: " << h << endl;
        return 0;
    }
    if (act == CHECK_GRAPH) {
        uint64_t h = 1469598103934665603ull;
        for (int32_t c : graph.col) { h ^= (uint32_t)c; h *= 1099511628211ull; }
        for (int64_t r : graph.row_ptr) { h ^= (uint64_t)r; h *= 1099511628211ull; }
        cout << "nnz: " << graph.col.size() << " csr_fnv1a: " << h << endl;
        return 0;
    } else if (act == GEN_SS_QUERY) { // algo.h:498-509, seeded instead of rand()
        const string f = config.graph_location + "ssquery.txt";
        if (FILE *t = fopen(f.c_str(), "r")) { fclose(t); puts("ss query set exists"); return 0; }
        std::ofstream qf(f);
        uint64_t s = config.seed ? config.seed : 1;
        for (unsigned i = 0; i < config.query_size; i++) {
            s ^= s << 13; s ^= s >> 7; s ^= s << 17;
            qf << (int)(s % (uint64_t)graph.n) << "\n";
        }
        return 0;
    } else if (act == QUERY) rc = do_query(graph);
    else if (act == TOPK) rc = do_topk(graph);
    else if (act == GEN_EXACT_TOPK) rc = do_gen_exact_topk(graph);
    else if (act == BATCH_TOPK) rc = do_batch_topk(graph);
    else if (act == BUILD) { const double t0 = now_s(); rc = do_build(graph); timers.add(0, now_s() - t0); }
    if (rc) return rc;
    timers.show(); // fora.cpp:282
    if (act == QUERY || act == TOPK) { // fora.cpp:283-287
        string cmd;
        for (int i = 1; i < argc; i++) cmd += string(" ") + argv[i];
        const string p = save_json(config, result, timers, start_time, cmd);
        cout << "result written to " << p << endl;
    }
    return 0;
}
