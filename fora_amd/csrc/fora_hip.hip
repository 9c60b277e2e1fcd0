// fora_hip.hip -- C ABI (include/fora_hip.h) over the gfx950 kernels of fora_kernels.h.
//
// Owns all device state of one GPU: the CSR graph, the optional walk index, and a
// workspace of `batch` query slots.  The host side only sequences kernel launches;
// there is no CPU fallback: without a HIP device every entry point fails.
#include "fora_kernels.h"
#include "fora_team.h"
#include "../../include/fora_hip.h"

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

using namespace fora;

struct fora_ctx;
static int build_hub_copy(fora_ctx *c, const int64_t *row_ptr, const int32_t *col); // (defined below, beside set_graph)
static int build_quad_copies(fora_ctx *c);
namespace {

struct EvPair {
    hipEvent_t a, b;
    int kind; // 0 pop, 1 expand, 2 walk_alloc, 3 walk, 4 other, 5 batch, 6 accum, 7 walk accum, 8 round sweep, 9 tail, 10 team push
};

} // namespace

// Knobs of the engine.  Read ONCE per context (fora_hip_create: environment FORA_HIP_<NAME>, upper case) and changed
// afterwards only through fora_hip_set_option; nothing on a query path calls getenv.
struct Tunables {
    int64_t direct = 0;          // 1: the one-atomic-per-edge push (test reference)
    int64_t force_wide = 0;      // 1: wide bucket layout on small graphs too (tests)
    int64_t pass_bins = 0;       // bins handled per pass in the wide layout (0: by graph size)
    int64_t no_split = 0;        // 1: multi-pass graphs without the row-sorted copy / split offsets (tests)
    int64_t no_compact = 0;      // 1: no bit-packed walk copy (set_graph)
    int64_t walk_dg = 2;         // online walks over the degree-grouped copy (k_walk_dg): 0 never, 1 with one gather per walk for the endpoint's id, 2 results in bucket order; read by set_graph and at launch
    int64_t hubs = -1;           // (-1: 1024, or 4096 when the team push is this graph's default -- then the bin kernel does not run and the only reader of the hub copy is k_push_tail: 5.48 -> 4.89 ms per 1000 ws queries) narrow layout: increments for the `hubs` nodes of largest in-degree are summed per workgroup in LDS (Dev::col_hub); 0: off; read by set_graph.  ws, push of 1000 queries: 0 -> 79.5 ms, 1024 -> 75.6, 2048 -> 83.2, 4096 -> 90.5 (the LDS table costs the bin kernel its occupancy; the accumulate is bound by its sweep, not by its messages)
    int64_t hubs_wide = -1;      // the same for graphs that run the wide layout in one pass per level; -1: 2048 up to 2^28 edges, else 0 (off); read by set_graph.
                                 // LJ-sized push of 280 queries: 0 -> 560.8 ms, 1024 -> 558.7, 2048 -> 550.6, 4096 -> 681.6; Twitter-2010-sized: no gain (the top 2048 of 41.6 M nodes receive few of the edges)
    int64_t hub_min = 4096;      // ... in levels whose frontier holds at least this many nodes of the slot
    int64_t dg_hubs = 0;         // hub records of that copy (0: the fewest that leave <= 255 degree classes); read by set_graph
    int64_t bkcap = 0;           // bucket capacity in messages (0: default per layout)
    int64_t ovcap = 0;           // overflow list capacity (0: scales with the graph)
    int64_t tiny = 512;          // k_accum: buckets up to this many messages go by direct atomics
    int64_t xb = 0, ax = 0, wx = 0; // workgroups per slot: bin kernel / slab sweeps / walks (0: from the slot count)
    int64_t tail = -1;           // frontier size (largest slot) from which k_push_tail takes over (0: never, -1: by slot count)
    int64_t tail_always = 0;     // 1: do not wait for the frontier to have been large first (tests)
    int64_t select_compact = -1; // top-k select over compacted non-zeros: -1 by graph size, 0 never, 1 always
    int64_t rounds = 1;          // threshold rounds of the bucketed push (k_round_sweep): 2^(rounds-1) x the threshold first; 1: plain.
                                 // 2 rounds relax 17 % fewer edges (ws) but need 99 instead of 61 level launches: push 88 -> 126 ms per 1000 queries
    int64_t round_div = 4;       // leave a threshold round once the frontier is down to 1/round_div of the round's largest (0: when it is empty)
    int64_t defer = 0;           // bounded deferral of the bucketed push (Dev::defer_k): a node that crosses with less than 2^defer x its threshold waits one level; 0 (default): plain levels.
                                 // CHANGES THE SCHEDULE (like rounds / round_div): results equal the twin run with the same value (orc_twin_set_defer).
                                 // ws, 1000 queries, defer 1: 13.5 % fewer relaxations and 6 % fewer walks, but 22-32 instead of 19 level launches and a longer tail: push 75 -> 99-114 ms
    int64_t defer_min = 0;       // with defer: only levels that pop at least this many nodes of the slot defer (orc_twin_set_defer_min)
    int64_t team = -1;           // graphs of the narrow layout push with k_push_team (a slot's residue resident in the LDS of a team of workgroups, fora_team.h): 0 never (the bucketed
                                 // kernels), 1 / -1 (default) always.  Same bits either way.  Measured, push of 1000 queries (round 6): ws-sized graph 45.7 + 4.9 ms against 77.8 bucketed; R-MAT
                                 // variant with 52 % dangling nodes (483 sources with out-edges) 24.3 + 5.4 against 41.5 (round 4: 47.3 against 41.3, hence a gate on the dangling share until round 6)
    int64_t team_size = 0;       // members per team (a power of two up to 32); 0: the fewest whose LDS holds the graph; read by set_graph
    int64_t team_tail = -1;      // frontier size (of a slot) at which k_push_team hands the slot to k_push_tail; 0: never; -1: 4096
    int64_t team_xcd = 1;        // 1: the members of a team share blockIdx % 8 (one XCD under round-robin placement: speed only)
    int64_t team_max = 0;        // teams per launch at most (0: one member per CU); tests
    int64_t team_hubs = 1024;    // k_push_team: increments for the nodes of largest in-degree are summed per member in LDS, one message per hub and level (0: off); read when the team tables are built
    int64_t tail_hubs = 1;       // k_push_tail: increments for the hubs of the hub copy are summed in LDS (0: every relaxation is an atomic)
    int64_t team_log = -1;       // k_push_team: entries of a member's reserve log per slot (-1: 2^17; 0: none, every pop adds to its accumulator; tests use small values for the mixed case)
    int64_t slot_major = -1;     // wide layouts: which launches put the slot in blockIdx.x (Dev::slot_major; bits 1 bin kernel, 2 accumulate, 4 indexed walks, 8 walk allocation); -1: by call type and slot count (make_dev); 0: round 5's order
    int64_t acc_group = 0;       // wide accumulate: bins per workgroup (0: by the launch's size, 1 ... 8; tests force 1 / 3 / 16)
    int64_t team_abort_level = 0; // tests: every team abandons its launch (as after a time-out) when a slot reaches this level -- an abort in mid-flight: partial slabs, logs, message buffers, tagged words
    int64_t team_timeout_ms = 500; // k_push_team: a member that has waited this long for its team gives up; the call then runs again through the bucketed kernels (with_retry)
    int64_t team_coop = 0;       // k_push_team launch: 0 (default) plain launch behind an occupancy check (occupancy x CUs >= grid, team_fits); 1: hipLaunchCooperativeKernel.  Measured on ROCm 7.2 / MI355X (round 5): the cooperative
                                 // launch costs ~9 ms per launch (push of 64 ws-sized queries 14.8 ms against 5.6: the runtime moves the launch to its cooperative queue and back) and a process with two contexts that used it crashed in the runtime's teardown -- opt-in only
    int64_t topk_bk_div = 16;    // top-k (--opt driver) on wide graphs: message buckets of 1 / this of a query's capacity (plan_workspace); 1: as large as a query's.
                                 // Twitter-2010-sized, k = 500 --opt --with_idx, 125 sources: 1 -> 245 q/s (8 slots per batch), 8 -> 279 (30), 16 -> 303 (37), 32 -> 304 (41), 64 -> 306 (44); same bits
    int64_t quads = 1;           // wide layouts, one bin pass per level: k_pushq_bin reads quad-padded copies of col / col_hub with 16-byte loads (0: single edges, round 4)
    int64_t pipeline = 0;        // 1: second lane (stream + workspace) when a call has more than one batch
    int64_t profile = 1;         // 0: no HIP event pairs around the launches
    int64_t grid = 2048;         // workgroups of the direct-path kernels
};
static const struct { const char *name; int64_t Tunables::*field; bool layout; } OPTIONS[] = {
    {"direct", &Tunables::direct, true}, {"force_wide", &Tunables::force_wide, true}, {"pass_bins", &Tunables::pass_bins, true},
    {"no_split", &Tunables::no_split, true}, {"no_compact", &Tunables::no_compact, false}, {"walk_dg", &Tunables::walk_dg, false}, {"dg_hubs", &Tunables::dg_hubs, false}, {"hubs", &Tunables::hubs, true}, {"hubs_wide", &Tunables::hubs_wide, true}, {"hub_min", &Tunables::hub_min, false}, {"bkcap", &Tunables::bkcap, true},
    {"ovcap", &Tunables::ovcap, true}, {"tiny", &Tunables::tiny, false}, {"xb", &Tunables::xb, false}, {"ax", &Tunables::ax, false},
    {"wx", &Tunables::wx, false}, {"tail", &Tunables::tail, false}, {"tail_always", &Tunables::tail_always, false},
    {"select_compact", &Tunables::select_compact, false}, {"pipeline", &Tunables::pipeline, false}, {"team", &Tunables::team, true}, {"team_size", &Tunables::team_size, true}, {"team_tail", &Tunables::team_tail, false}, {"team_xcd", &Tunables::team_xcd, false}, {"team_max", &Tunables::team_max, true}, {"team_hubs", &Tunables::team_hubs, true}, {"team_log", &Tunables::team_log, false}, {"topk_bk_div", &Tunables::topk_bk_div, true}, {"quads", &Tunables::quads, false}, {"team_timeout_ms", &Tunables::team_timeout_ms, false}, {"team_abort_level", &Tunables::team_abort_level, false}, {"acc_group", &Tunables::acc_group, false}, {"slot_major", &Tunables::slot_major, false}, {"team_coop", &Tunables::team_coop, false}, {"tail_hubs", &Tunables::tail_hubs, false}, {"rounds", &Tunables::rounds, false}, {"defer", &Tunables::defer, true}, {"defer_min", &Tunables::defer_min, false}, {"round_div", &Tunables::round_div, false},
    {"profile", &Tunables::profile, false}, {"grid", &Tunables::grid, false},
};
// knobs that choose another push SCHEDULE (other, equally valid result bits): never taken from the environment -- a stray
// variable must not change what a query returns; fora_hip_set_option sets them (tests, experiments)
static bool schedule_option(const char *name) {
    return !strcmp(name, "rounds") || !strcmp(name, "round_div") || !strcmp(name, "defer") || !strcmp(name, "defer_min");
}
static Tunables tunables_from_env() {
    Tunables t;
    for (const auto &o : OPTIONS) {
        if (schedule_option(o.name)) continue;
        std::string env = "FORA_HIP_";
        for (const char *p = o.name; *p; p++) env += (char)toupper((unsigned char)*p);
        if (const char *e = getenv(env.c_str())) if (*e) t.*(o.field) = atoll(e);
    }
    return t;
}

struct fora_ctx {
    int device = 0;
    Tunables opt_;
    hipStream_t stream = nullptr;
    std::string err;
    hipDeviceProp_t prop{};

    // graph
    int32_t n = 0;
    int64_t m_attr = 0, nnz = 0;
    std::vector<int64_t> h_row_ptr;
    int64_t *d_row_ptr = nullptr;
    int32_t *d_col = nullptr;
    uint64_t *d_rowinfo = nullptr;
    uint32_t *d_deg = nullptr;
    uint32_t *d_rp32 = nullptr, *d_colp = nullptr;
    int32_t *d_col_push = nullptr;  // row-sorted copy of col (multi-pass graphs whose rows are not sorted)
    uint32_t *d_row_split = nullptr; // [n][npass + 1]
    int split_pbins = 0;
    uint32_t colbits = 0;
    // degree-grouped walk copy (WalkDG): device arrays + the scalars of the struct; dg.colp == nullptr: none
    int32_t *d_col_hub = nullptr;    // hub pre-aggregation (Dev::col_hub)
    int32_t *d_col4 = nullptr, *d_col_hub4 = nullptr; // quad-padded copies for the wide bin kernel (Dev::col4); null: not built
    uint64_t *d_rowinfo4 = nullptr;
    uint64_t quads = 0;              // quads of the padded copies
    uint32_t *d_hub_node = nullptr, *d_hub_first = nullptr;
    uint32_t hubs = 0;
    int hub_shift = 0;               // bin shift the hub ranges were built for
    uint64_t *d_hubsum = nullptr;    // workspace: [B][sub][hubs]
    uint32_t *d_dg_perm = nullptr, *d_dg_inv = nullptr, *d_dg_colp = nullptr, *d_dg_rec = nullptr, *d_dg_invb = nullptr;
    uint8_t *d_dg_T = nullptr;
    WalkDG dg{};
    // team push (fora_team.h): target copy of col, bucket offsets; built by set_graph for graphs of the narrow layout
    uint32_t *d_colt = nullptr, *d_team_off = nullptr, *d_team_n2l = nullptr, *d_team_l2n = nullptr;
    uint32_t *d_team_hubtgt = nullptr;
    uint32_t team_H = 0, team_hubs_opt = 0;
    uint16_t *d_team_rlog_id = nullptr; uint64_t *d_team_rlog_val = nullptr; uint32_t team_rlog_cap = 0; // reserve logs (TeamDev::rlog_id)
    uint64_t *d_team_rowl = nullptr, *d_team_rsvl = nullptr; // rows by local id (graph); reserve accumulators by local id (workspace)
    uint16_t *d_team_deg16 = nullptr;
    uint32_t *d_team_rowq = nullptr; // [n] first quad of every node's row in d_colt
    uint32_t team_T = 0, team_R = 0, team_force = 0; // members per team, local ids per member; the team_size option they were built for
    bool team_checked = false, team_wanted = false;  // ensure_team has looked at this graph with these options
    double dangling_frac = 0;        // share of the nodes without out-edges
    uint64_t team_cap = 0;           // message slots per (team, parity)
    // ... and its workspace
    uint32_t *d_team_msg = nullptr;
    uint64_t *d_team_inct = nullptr;
    unsigned long long *d_team_cnt = nullptr; // the teams' barrier words (TeamDev::cntw)
    uint32_t *d_team_ctl = nullptr; // ctl: [0] next slot, [32] abort | sync words | slot sequences
    uint32_t team_n = 0;             // teams of a launch
    bool team_dirty = false;         // a launch ended with an error flag: its reserve accumulators (TeamDev::rsvl) may not be zero
    bool hub_for_team = false;       // the hub copy was sized for the team path (4096 hubs, k_push_tail its only reader)
    bool team_timeout_seen = false;  // the last device error was ERR_TEAM_TIMEOUT (with_retry runs the call again without the team push)
    int team_suspend = 0;            // calls left that push with the bucketed kernels after a team time-out
    uint64_t team_fallbacks = 0;     // calls re-run that way so far (fora_hip_get_option "team_fallbacks")
    int team_fit = -1;               // 1: every workgroup of a k_push_team launch fits the device at once (occupancy x CUs >= grid); 0: no team push; -1: not asked yet
    bool team_coop_ok = false;       // launch k_push_team cooperatively (option team_coop, device attribute)
    bool team_coop_failed = false;   // hipLaunchCooperativeKernel refused once: plain launches from then on

    // params
    bool have_params = false;
    double alpha = 0.2, epsilon = 0.5, rmax_scale = 1.0, rmax = 0, omega = 0;
    int opt = 0;
    uint64_t seed = 0;

    // index
    int32_t *d_rw_idx = nullptr;
    uint64_t *d_idx_off = nullptr, *d_idx_cnt = nullptr;
    uint64_t idx_len = 0;
    bool have_index = false;

    // workspace
    int batch_req = 0, B = 0;
    int B_memcap = 0; // slots that fitted the free memory when the workspace was planned (>= B)
    uint64_t *d_residue = nullptr, *d_ppr = nullptr, *d_wl[2] = {nullptr, nullptr};
    void *d_scratch = nullptr; // PushSeg list during the push, WalkItem list during the walks
    uint64_t wl_cap = 0, seg_cap = 0, wit_cap = 0;
    unsigned long long *d_counters = nullptr; // wl_count | seg_count | wit_count | tot_steps
    QState *d_qs = nullptr;
    int32_t *d_src = nullptr;
    uint32_t *d_err = nullptr;
    unsigned long long *d_stamps = nullptr; // diagnostic builds (-DFORA_STAMPS)
    // bucketed push (n <= MAX_BINS * BIN_SIZE)
    bool binned = false;
    int nbins = 0, pbins = 0; // bins of the graph; bins per pass (bucket-array stride)
    uint32_t *d_fl[2] = {nullptr, nullptr}, *d_fl_count = nullptr; // fl_count: [2][B]
    uint64_t *d_inc_tab[2] = {nullptr, nullptr};
    uint32_t *d_ov_w = nullptr, *d_ov_count = nullptr, *d_ov_bin = nullptr; // bucket overflow list, its size, its entries per bin [2][B][nbins]
    uint64_t *d_ov_inc = nullptr;
    uint32_t ov_cap = 0;
    uint32_t *d_bk_w = nullptr, *d_bk_count = nullptr;
    uint64_t *d_bk_inc = nullptr;
    uint64_t segq_cap = 0;
    uint32_t bk_cap = 0, sub = 0; // capacity of one sub-bucket; sub-buckets per (slot, bin) = producer workgroups per slot
    uint32_t *d_wit_count = nullptr; // [B * CSTRIDE]
    uint32_t *d_sw = nullptr;        // [2][B * CSTRIDE] k_round_sweep: append counters, finished-workgroup tickets
    uint32_t *d_tile_ctr = nullptr;  // [2][B * CSTRIDE] wide bin kernels: next tile of a slot (Dev::tile_ctr)
    uint64_t *d_dbm = nullptr;       // [2][B][dbm_words] bounded deferral: marks of the deferred nodes (Dev::dbm)
    uint32_t *d_dflag = nullptr;     // [2][B][nbins]
    uint32_t *d_dl = nullptr;        // [2][B][n] k_push_tail's deferred lists
    uint32_t dbm_words = 0;
    uint64_t bin_launches = 0;       // parity picks the counter set
    uint32_t *h_flc = nullptr; // pinned ring of per-slot frontier sizes
    uint64_t *d_ppr2 = nullptr, *d_cursor = nullptr; // top-k: per-round ppr, index cursors (rw_counter)
    uint32_t cursor_epoch = 0;       // batch serial number stamped into the cursor words (k_walk_alloc): 0 = the slabs hold no valid word
    uint8_t *d_active = nullptr;
    unsigned long long *d_above = nullptr;
    double *d_sel_thr = nullptr; // [B] launch_select: per-slot limit of the entries that can be among the top k
    // top-k with bounds: upper_bounds / lower_bounds (query.h:1350-1353), topk_filter marks, stop flags, walks of the round
    double *d_upper = nullptr, *d_lower = nullptr;
    uint8_t *d_filter = nullptr;
    uint32_t *d_fail = nullptr;
    unsigned long long *d_round_walks = nullptr;
    uint32_t *d_nz_counts = nullptr; // [B][NZ_X + 1]: per-block non-zero counts, then the slot's total
    double *d_lb_sc = nullptr;
    int32_t *d_lb_ids = nullptr;
    int lb_cap = 0;
    int32_t *d_topk_ids = nullptr;
    double *d_topk_sc = nullptr;
    int topk_cap = 0;
    unsigned long long *h_pinned = nullptr; // [MAX_LEVELS + 2] frontier sizes read back
    std::vector<QState> h_qs;
    QState *h_qs_pin = nullptr;              // pinned landing area of the per-slot accumulators
    unsigned long long *h_steps_pin = nullptr;
    // second lane: own stream + workspace, shares graph / index / params; lets the push of batch
    // k+1 overlap the (fabric-bound) walks of batch k
    fora_ctx *twin = nullptr;
    bool is_twin = false;
    uint32_t bk_scale = 1;        // bucket capacity multiplier, doubled after a bucket overflow (see with_bucket_retry)
    uint32_t bk_scale_topk = 1;   // ... of the calls that plan with a divisor (bk_div > 1: the top-k driver on wide graphs).  Its own word: those calls start at 1 / 16 of a
                                  // query's buckets and overflow far more often; a doubling there must not shrink the batches of later query / power-iteration calls
    uint64_t bucket_retries = 0;  // calls re-run with doubled buckets so far (fora_hip_get_option "bucket_retries")
    uint32_t bk_div = 1;          // bucket capacity divisor of the call in progress (top-k: TOPK_BK_DIV on wide graphs, plan_workspace)
    bool bucket_overflow = false; // the last device error was ERR_BUCKET_OVERFLOW
    // --balanced (query.h:848-884): cost model in seconds
    bool balanced = false;
    double c_pop = 2.0e-11, c_edge = 2.4e-11, t_walk = 6.5e-11, t_idx = 2.2e-11, bal_start = 8;
    std::vector<double> h_rmax_used;
    std::vector<int32_t> h_rounds;
    int pending_nq = 0;                      // batch enqueued on this lane, not yet finished

    // timing
    bool profiling = true;
    std::vector<EvPair> ev_pool;
    size_t ev_used = 0;
    fora_timing timing{};
    int grid_blocks = 2048;
};

namespace {

int fail(fora_ctx *c, int code, const std::string &msg) {
    if (c) c->err = msg;
    return code;
}

#define HIPCHK(c, call)                                                                          \
    do {                                                                                         \
        hipError_t e_ = (call);                                                                  \
        if (e_ != hipSuccess)                                                                    \
            return fail(c, e_ == hipErrorOutOfMemory ? FORA_E_NOMEM : FORA_E_HIP,               \
                        std::string(#call) + ": " + hipGetErrorString(e_));                      \
    } while (0)

template <typename T> void dfree(T *&p) {
    if (p) (void)hipFree(p);
    p = nullptr;
}

void free_graph(fora_ctx *c) {
    dfree(c->d_row_ptr); dfree(c->d_col); dfree(c->d_rowinfo); dfree(c->d_deg); dfree(c->d_rp32); dfree(c->d_colp); dfree(c->d_col_push); dfree(c->d_row_split);
    dfree(c->d_col_hub); dfree(c->d_hub_node); dfree(c->d_hub_first); c->hubs = 0;
    dfree(c->d_col4); dfree(c->d_col_hub4); dfree(c->d_rowinfo4); c->quads = 0;
    dfree(c->d_colt); dfree(c->d_team_rowq); dfree(c->d_team_off); dfree(c->d_team_n2l); dfree(c->d_team_l2n); dfree(c->d_team_deg16); dfree(c->d_team_rowl); dfree(c->d_team_hubtgt); c->team_H = 0; c->team_T = 0; c->team_R = 0; c->team_cap = 0; c->team_checked = false;
    dfree(c->d_dg_perm); dfree(c->d_dg_inv); dfree(c->d_dg_colp); dfree(c->d_dg_rec); dfree(c->d_dg_T); dfree(c->d_dg_invb);
    c->dg = WalkDG{};
    c->split_pbins = 0;
    c->n = 0; c->nnz = 0;
}
void free_index(fora_ctx *c) {
    dfree(c->d_rw_idx); dfree(c->d_idx_off); dfree(c->d_idx_cnt);
    c->idx_len = 0; c->have_index = false;
}
void free_workspace(fora_ctx *c) {
    dfree(c->d_residue); dfree(c->d_ppr); dfree(c->d_wl[0]); dfree(c->d_wl[1]); dfree(c->d_scratch);
    dfree(c->d_counters); dfree(c->d_qs); dfree(c->d_src); dfree(c->d_err);
    dfree(c->d_ppr2); dfree(c->d_cursor); c->cursor_epoch = 0; dfree(c->d_active); dfree(c->d_above); dfree(c->d_sel_thr); dfree(c->d_topk_ids); dfree(c->d_topk_sc);
    dfree(c->d_upper); dfree(c->d_lower); dfree(c->d_filter); dfree(c->d_fail); dfree(c->d_round_walks);
    dfree(c->d_lb_sc); dfree(c->d_lb_ids); c->lb_cap = 0;
    dfree(c->d_nz_counts);
    c->topk_cap = 0;
    dfree(c->d_fl[0]); dfree(c->d_fl[1]); dfree(c->d_fl_count); dfree(c->d_inc_tab[0]); dfree(c->d_inc_tab[1]); dfree(c->d_ov_w); dfree(c->d_ov_inc); dfree(c->d_ov_count); dfree(c->d_ov_bin);
    dfree(c->d_bk_w); dfree(c->d_bk_inc); dfree(c->d_bk_count); dfree(c->d_wit_count); dfree(c->d_sw); dfree(c->d_tile_ctr);
    dfree(c->d_dbm); dfree(c->d_dflag); dfree(c->d_dl); dfree(c->d_hubsum);
    dfree(c->d_team_msg); dfree(c->d_team_inct); dfree(c->d_team_rsvl); dfree(c->d_team_rlog_id); dfree(c->d_team_rlog_val); dfree(c->d_team_cnt); dfree(c->d_team_ctl); c->team_n = 0; c->team_fit = -1;
    if (c->h_flc) (void)hipHostFree(c->h_flc);
    c->h_flc = nullptr;
    if (c->h_pinned) (void)hipHostFree(c->h_pinned);
    c->h_pinned = nullptr;
    if (c->h_qs_pin) (void)hipHostFree(c->h_qs_pin);
    c->h_qs_pin = nullptr;
    if (c->h_steps_pin) (void)hipHostFree(c->h_steps_pin);
    c->h_steps_pin = nullptr;
    c->B = 0;
    c->B_memcap = 0;
}

constexpr size_t N_COUNTERS = 2 * (size_t)(MAX_LEVELS + 2) + 2;
// workgroups per slot of the kernels that sweep a slot's slab (walk allocation, top-k frontier / copy / count): ~32 k
// workgroups per launch; one per 256 nodes (up to 1 M tiny workgroups at 1000 slots) cost k_walk_alloc 19 ms instead of
// 8 per 3000 ws queries
static uint32_t slab_grid_x(const fora_ctx *c, int nq) {
    const int64_t nchunk = ((int64_t)c->n + BLOCK - 1) / BLOCK;
    int64_t x = std::min<int64_t>(1024, std::max<int64_t>(16, 32768 / std::max(1, nq)));
    if (c->opt_.ax > 0) x = c->opt_.ax;
    return (uint32_t)std::max<int64_t>(1, std::min<int64_t>(std::min(x, nchunk), 65535)); // (may be a grid's y extent: Dev::slot_major)
}
static unsigned walk_grid_x(const fora_ctx *c, int nq) {
    if (c->opt_.wx > 0) return (unsigned)c->opt_.wx;
    // ~24 k workgroups per launch (1280 are resident): ws at 1000 slots, blocks per slot 4 -> 552 ms, 8 -> 509,
    // 16 -> 493, 24 -> 489, 32 -> 495 per 3000 queries
    return (unsigned)std::min(2048, std::max(16, 24576 / std::max(1, nq)));
}

constexpr int SPEC = 3;          // levels launched ahead of the frontier-size readback
constexpr int FLC_RING = SPEC + 2;

static bool want_binned(const fora_ctx *c) { return c->opt_.direct != 1; } // direct: the one-atomic-per-edge path (tests)
// bins handled per pass in the wide layout (graphs with more bins run several bin/accum passes per level)
static int want_pass_bins(const fora_ctx *c, int nbins) {
    // graphs with more than 1024 bins run several passes per level; up to 2560 bins per pass then (the per-pass
    // re-scan of the frontier and of the walk index costs more than the shorter message runs: Twitter-2010-sized,
    // 26 queries, bins per pass 256 / 512 / 1024: 7.5 / 4.3 / 3.0 s)
    const int cap = nbins > MAX_BINS_WIDE ? MAX_BINS_HUGE : MAX_BINS_WIDE;
    if (c->opt_.pass_bins > 0) return (int)std::min<int64_t>(c->opt_.pass_bins, cap);
    if (nbins > cap) { const int np = (nbins + cap - 1) / cap; return (nbins + np - 1) / np; } // equal passes
    return cap;
}
// narrow layout: <= MAX_BINS bins and the slice index fits the 4-byte push message
static bool want_wide(const fora_ctx *c) {
    if (c->opt_.force_wide == 1) return true; // tests: exercise the wide layout on small graphs
    return !((uint64_t)c->n <= (uint64_t)MAX_BINS * BIN_SIZE && (uint64_t)c->n <= (1ull << SEG_BITS));
}
// bits of a node id inside its bin: 8192-node bins in the narrow layout, 16384 in the wide ones
static int bin_shift(const fora_ctx *c) { return want_wide(c) ? BIN_SHIFT_WIDE : BIN_SHIFT; }
static uint64_t bins_of(const fora_ctx *c) { const int sh = bin_shift(c); return ((uint64_t)c->n + (1ull << sh) - 1) >> sh; }
static uint32_t want_bk_cap(const fora_ctx *c) {
    if (c->opt_.bkcap > 0) return (uint32_t)c->opt_.bkcap;
    return 163840; // walk results: ~omega*rsum/nbins per bucket (ws: ~110 k)
}
static uint32_t want_bk_cap_wide(const fora_ctx *c) { // messages per (slot, bin) bucket: push increments and indexed walk results (online ones go by direct atomics)
    if (c->opt_.bkcap > 0) return (uint32_t)c->opt_.bkcap;
    // a dense level relaxes about every edge once: nnz / nbins messages per bin on average (Twitter-2010-sized: 289 k),
    // hubs' bins beyond that use the overflow list; 196608 covers the indexed walk results (~omega*rsum/nbins per bucket)
    const uint64_t nbins = bins_of(c);
    return (uint32_t)std::min<uint64_t>(std::max<uint64_t>(196608, (uint64_t)(1.4 * (double)c->nnz / (double)std::max<uint64_t>(1, nbins))), 1u << 26);
}

struct WsPlan { uint64_t segs, wits, scratch, per_slot; int nbins, pbins; uint32_t bk_cap, sub; uint64_t segq_cap; bool binned; };
// (the wide / narrow choice changes bk_cap, which forces a re-plan of the workspace)
// sub-buckets per (slot, bin) = producer workgroups per slot (Dev::bk_w): ~16 k producer workgroups per launch
static uint32_t want_sub(const fora_ctx *c, int slots) {
    if (c->opt_.xb > 0) return (uint32_t)std::min<int64_t>(c->opt_.xb, MAX_SUB);
    // wide: 512-thread producers, 2-3 resident per CU.  LJ-sized, 74 slots: 16 -> 549 ms per 148 queries, 32 -> 492, 64 -> 515,
    // 128 -> 539; Twitter-sized, 12 slots, 24 queries: 32 -> 2438 ms, 64 -> 2010, 128 -> 1576 (few slots: the tiles of a level
    // have to be dealt to many workgroups)
    // (16384-node bins, LJ-sized, 140 slots, 280 queries: 16 -> 902 ms, 32 -> 840, 64 -> 826)
    if (want_wide(c)) return slots >= 256 ? 32u : slots >= 32 ? 64u : (uint32_t)MAX_SUB;
    return (uint32_t)std::min(MAX_SUB, std::max(16, 16384 / std::max(1, slots)));
}
static WsPlan plan_workspace(const fora_ctx *c, double omega_hint, int slots) {
    WsPlan p{};
    const uint64_t n = (uint64_t)c->n;
    p.binned = want_binned(c);
    p.segs = n + (uint64_t)c->nnz / PUSH_SEG + 64; // per slot
    double walks = omega_hint > 0 ? omega_hint : 0;
    if (walks > 4e12) walks = 4e12;
    p.wits = n + n / WALK_SEG + (uint64_t)(walks / WALK_SEG) + 64;
    if (p.binned) {
        p.nbins = (int)bins_of(c);
        p.pbins = want_wide(c) ? std::min(p.nbins, want_pass_bins(c, p.nbins)) : std::max(p.nbins, (int)c->dg.nbx); // narrow: the walk results in bucket order may need a bin more
        p.sub = want_sub(c, slots);
        { // capacity of one sub-bucket: the bucket's capacity over its sub-buckets (+25 % for uneven producers); the
          // `bkcap` option (tests) sets it directly
            const uint64_t total = (uint64_t)(want_wide(c) ? want_bk_cap_wide(c) : want_bk_cap(c));
            uint64_t cap = c->opt_.bkcap > 0 ? total : (total + total / 4 + p.sub - 1) / p.sub;
            // the top-k driver's rounds push from small frontiers (delta starts at 1 / 10k): its buckets start at 1 / bk_div of
            // a query's -- a slot is a fifth of the memory, a batch holds that many more of them, and every per-round launch
            // (k_push_tail: ONE workgroup per slot; the slab sweeps; the walk kernels) works on that many more slots at once.
            // A round that does overflow is run again with doubled buckets like any other (with_bucket_retry)
            cap = std::min<uint64_t>(std::max<uint64_t>(cap * (c->bk_div > 1 ? c->bk_scale_topk : c->bk_scale) / std::max<uint32_t>(1, c->bk_div), 64), 1u << 28);
            p.bk_cap = (uint32_t)((cap + 15) & ~15ull);
        }
        p.segq_cap = n; // frontier positions
        p.scratch = (p.wits * sizeof(WalkItemP) + 95) / 96 * 96; // whole PushSeg (24 B) and WalkItemP (32 B) entries: `keepable` compares seg_cap * sizeof(PushSeg) with it
        p.per_slot = n * 8 * 2 + n * 4 * 2 + p.segq_cap * 8 * 2 + std::max<uint64_t>(262144, n / 8) * 12 + (uint64_t)p.pbins * p.sub * p.bk_cap * (want_wide(c) ? 8 : 12) + p.scratch +
                     (c->opt_.defer > 0 ? n * 4 * 2 : 0) + n / 4 + 64 + (uint64_t)p.sub * c->hubs * 8; // + deferred lists and bitmaps, hub sums
    } else {
        p.scratch = (std::max(p.segs * sizeof(PushSeg), p.wits * sizeof(WalkItemP)) + 95) / 96 * 96;
        p.per_slot = n * 8 * 4 + p.scratch;
    }
    return p;
}

// Multi-pass graphs (more bins than one pass holds): row-sorted copy of col + per-row split offsets, so that every
// pass of k_pushq_bin reads only its own part of each popped row.  Built once per (graph, pass size).
int ensure_row_split(fora_ctx *c, int nbins, int pbins) {
    const int npass = pbins > 0 ? (nbins + pbins - 1) / pbins : 1;
    if (c->is_twin) return FORA_OK; // shares the first lane's tables (sync_twin)
    if (npass <= 1 || c->opt_.no_split) { dfree(c->d_col_push); dfree(c->d_row_split); c->split_pbins = 0; return FORA_OK; }
    if (c->d_row_split && c->split_pbins == pbins) return FORA_OK;
    dfree(c->d_col_push); dfree(c->d_row_split);
    const size_t n = (size_t)c->n, nnz = (size_t)c->nnz;
    std::vector<int32_t> col(std::max<size_t>(1, nnz));
    if (nnz) HIPCHK(c, hipMemcpy(col.data(), c->d_col, nnz * 4, hipMemcpyDeviceToHost));
    bool sorted = true;
    for (size_t v = 0; v < n && sorted; v++)
        for (int64_t e = c->h_row_ptr[v] + 1; e < c->h_row_ptr[v + 1]; e++)
            if (col[(size_t)e - 1] > col[(size_t)e]) { sorted = false; break; }
    if (!sorted) {
        for (size_t v = 0; v < n; v++) std::sort(col.begin() + c->h_row_ptr[v], col.begin() + c->h_row_ptr[v + 1]);
        HIPCHK(c, hipMalloc(&c->d_col_push, std::max<size_t>(1, nnz) * 4));
        HIPCHK(c, hipMemcpy(c->d_col_push, col.data(), nnz * 4, hipMemcpyHostToDevice));
    }
    std::vector<uint32_t> split(n * (size_t)(npass + 1));
    for (size_t v = 0; v < n; v++) {
        const int32_t *rb = col.data() + c->h_row_ptr[v], *re = col.data() + c->h_row_ptr[v + 1];
        uint32_t *sp = split.data() + v * (size_t)(npass + 1);
        sp[0] = 0;
        for (int p = 1; p < npass; p++) {
            const int64_t first_node = ((int64_t)p * pbins) << bin_shift(c);
            sp[p] = (uint32_t)(std::lower_bound(rb, re, (int32_t)std::min<int64_t>(first_node, INT32_MAX)) - rb);
        }
        sp[npass] = (uint32_t)(re - rb);
    }
    HIPCHK(c, hipMalloc(&c->d_row_split, split.size() * 4));
    HIPCHK(c, hipMemcpy(c->d_row_split, split.data(), split.size() * 4, hipMemcpyHostToDevice));
    c->split_pbins = pbins;
    return FORA_OK;
}

// Team push (fora_team.h): members per team for this graph (0: the graph does not take the team path), the copy of
// col that names every target as (owner, local id) and the exact bucket capacities.  Built on first use and whenever
// the `team` / `team_size` options ask for another shape.
static bool want_team(const fora_ctx *c) {
    const bool on = c->opt_.team != 0; // (round 6: also for graphs with many dangling nodes -- 29.7 against 41.7 ms on the R-MAT ws variant, profiles/r06_dangling_team.txt)
    return on && want_binned(c) && !want_wide(c) && c->nnz > 0 && c->nnz < (1ll << 32); // (rowl / colt / off index edges with 32 bits)
}
int ensure_team(fora_ctx *c) {
    if (c->is_twin) return FORA_OK; // shares the first lane's tables (sync_twin)
    const bool want = want_team(c);
    const uint32_t force = (uint32_t)std::min<int64_t>(std::max<int64_t>(c->opt_.team_size, 0), TEAM_MAX);
    const uint32_t hubs_opt = (uint32_t)std::min<int64_t>(std::max<int64_t>(c->opt_.team_hubs, 0), 4096);
    if (c->team_checked && want == c->team_wanted && (!want || (c->team_force == force && c->team_hubs_opt == hubs_opt))) return FORA_OK;
    c->team_checked = true; c->team_wanted = want;
    dfree(c->d_colt); dfree(c->d_team_rowq); dfree(c->d_team_off); dfree(c->d_team_n2l); dfree(c->d_team_l2n); dfree(c->d_team_deg16); dfree(c->d_team_rowl); dfree(c->d_team_hubtgt);
    c->team_H = 0; c->team_hubs_opt = hubs_opt;
    c->team_T = 0; c->team_R = 0; c->team_cap = 0; c->team_force = force;
    if (!want) return FORA_OK;
    const size_t n = (size_t)c->n, nnz = (size_t)c->nnz;
    std::vector<int32_t> col(nnz);
    HIPCHK(c, hipMemcpy(col.data(), c->d_col, nnz * 4, hipMemcpyDeviceToHost));
    std::vector<uint32_t> indeg(n, 0);
    for (size_t e = 0; e < nnz; e++) indeg[(size_t)col[e]]++;
    // members per team: the fewest (a power of two) whose LDS holds their share of the nodes that have in-edges
    uint32_t T = 1;
    while (T < force) T *= 2;
    std::vector<uint32_t> cntm;
    uint32_t R = 0;
    for (;; T *= 2) {
        if (T > (uint32_t)TEAM_MAX || T > (uint32_t)std::max(1, c->prop.multiProcessorCount * TEAM_WGS_PER_CU)) return FORA_OK; // too large for the team path
        cntm.assign(T, 0);
        for (size_t v = 0; v < n; v++) if (indeg[v]) cntm[(v >> 6) % T]++;
        R = (*std::max_element(cntm.begin(), cntm.end()) + 63) / 64 * 64;
        if (R == 0) R = 64;
        if (R <= TEAM_R_CAP) break;
    }
    std::vector<uint32_t> n2l(n, TEAM_EMPTY), l2n((size_t)T * R, TEAM_EMPTY);
    std::vector<uint16_t> deg16((size_t)T * R, 0);
    std::vector<uint64_t> rowl((size_t)T * R, 0);
    std::fill(cntm.begin(), cntm.end(), 0);
    for (size_t v = 0; v < n; v++) {
        if (!indeg[v]) continue;
        const uint32_t s = (uint32_t)((v >> 6) % T), l = cntm[s]++;
        n2l[v] = (s << TEAM_LBITS) | l;
        l2n[(size_t)s * R + l] = (uint32_t)v;
        const int64_t dg = c->h_row_ptr[v + 1] - c->h_row_ptr[v];
        deg16[(size_t)s * R + l] = (uint16_t)std::min<int64_t>(dg, 0xFFFF);
        rowl[(size_t)s * R + l] = (uint64_t)v | ((uint64_t)std::min<int64_t>(dg, 8191) << 19); // n <= 2^19; the row's first quad (<< 32) follows below
    }
    // rows of the team copy are padded to whole quads (four words, 16-byte aligned): a lane reads a quad with one load
    std::vector<uint32_t> rowq(n + 1, 0);
    for (size_t v = 0; v < n; v++) rowq[v + 1] = rowq[v] + (uint32_t)((c->h_row_ptr[v + 1] - c->h_row_ptr[v] + 3) / 4); // (< 2^32: nnz < 2^32, want_team)
    for (size_t v = 0; v < n; v++)
        if (n2l[v] != TEAM_EMPTY) rowl[(size_t)(n2l[v] >> TEAM_LBITS) * R + (n2l[v] & TEAM_LMASK)] |= (uint64_t)rowq[v] << 32;
    // hubs: the nodes of largest in-degree (ties: lower id); their sums travel as one message per member and level
    // (their LDS sums share the 160 KiB with the residues and ~23 KB of static arrays)
    const uint64_t lds_left = 163840 / TEAM_WGS_PER_CU - 23 * 1024 - ((uint64_t)R + 1) * 8;
    const uint32_t Hn = (uint32_t)std::min<uint64_t>(std::min<uint64_t>(hubs_opt, n), lds_left / 8);
    std::vector<uint32_t> hub_of(n, TEAM_EMPTY), hubtgt(std::max<uint32_t>(1, Hn), 0);
    std::vector<uint8_t> hub_ok(std::max<uint32_t>(1, Hn), 0);
    if (Hn) {
        std::vector<uint32_t> order(n);
        for (size_t v = 0; v < n; v++) order[v] = (uint32_t)v;
        std::partial_sort(order.begin(), order.begin() + Hn, order.end(),
                          [&](uint32_t x, uint32_t y) { return indeg[x] != indeg[y] ? indeg[x] > indeg[y] : x < y; });
        for (uint32_t h = 0; h < Hn; h++) if (indeg[order[h]]) { hub_of[order[h]] = h; hubtgt[h] = n2l[order[h]]; hub_ok[h] = 1; }
    }
    std::vector<uint32_t> colt((size_t)rowq[n] * 4, TEAM_EMPTY);
    std::vector<uint64_t> pair((size_t)T * T, 0);
    for (size_t v = 0; v < n; v++) {
        const uint32_t s = (uint32_t)((v >> 6) % T);
        for (int64_t e = c->h_row_ptr[v]; e < c->h_row_ptr[v + 1]; e++) {
            const uint32_t t = (uint32_t)col[(size_t)e], w = n2l[t];
            colt[(size_t)rowq[v] * 4 + (size_t)(e - c->h_row_ptr[v])] = hub_of[t] != TEAM_EMPTY ? (0x80000000u | hub_of[t]) : w;
            if (hub_of[t] == TEAM_EMPTY) pair[(size_t)s * T + (w >> TEAM_LBITS)]++;
        }
    }
    for (uint32_t h = 0; h < Hn; h++) // a member sends a hub at most one message per level
        if (hub_ok[h]) for (uint32_t s = 0; s < T; s++) pair[(size_t)s * T + (hubtgt[h] >> TEAM_LBITS)]++;
    // bucket (s -> d): one 4-byte message per edge + the dangling mass of the level; whole 64-byte lines
    std::vector<uint32_t> off((size_t)T * T + 1, 0);
    uint64_t at = 0;
    for (size_t i = 0; i < (size_t)T * T; i++) {
        off[i] = (uint32_t)at;
        at += (pair[i] + 1 + 15) & ~15ull;
        if (at >= (1ull << 32) || pair[i] + 1 >= (1ull << 24)) return FORA_OK; // 32-bit slots; a bucket's count is 24 bits of its barrier word: no team push for such a graph
    }
    off[(size_t)T * T] = (uint32_t)at;
    HIPCHK(c, hipMalloc(&c->d_colt, colt.size() * 4 + 16));
    HIPCHK(c, hipMalloc(&c->d_team_rowq, n * 4));
    HIPCHK(c, hipMemcpy(c->d_team_rowq, rowq.data(), n * 4, hipMemcpyHostToDevice));
    HIPCHK(c, hipMalloc(&c->d_team_off, off.size() * 4));
    HIPCHK(c, hipMalloc(&c->d_team_n2l, n * 4));
    HIPCHK(c, hipMalloc(&c->d_team_l2n, l2n.size() * 4));
    HIPCHK(c, hipMalloc(&c->d_team_deg16, deg16.size() * 2));
    HIPCHK(c, hipMalloc(&c->d_team_hubtgt, hubtgt.size() * 4));
    HIPCHK(c, hipMemcpy(c->d_team_hubtgt, hubtgt.data(), hubtgt.size() * 4, hipMemcpyHostToDevice));
    c->team_H = Hn;
    HIPCHK(c, hipMalloc(&c->d_team_rowl, rowl.size() * 8));
    HIPCHK(c, hipMemcpy(c->d_team_rowl, rowl.data(), rowl.size() * 8, hipMemcpyHostToDevice));
    HIPCHK(c, hipMemcpy(c->d_colt, colt.data(), colt.size() * 4, hipMemcpyHostToDevice));
    HIPCHK(c, hipMemcpy(c->d_team_off, off.data(), off.size() * 4, hipMemcpyHostToDevice));
    HIPCHK(c, hipMemcpy(c->d_team_n2l, n2l.data(), n * 4, hipMemcpyHostToDevice));
    HIPCHK(c, hipMemcpy(c->d_team_l2n, l2n.data(), l2n.size() * 4, hipMemcpyHostToDevice));
    HIPCHK(c, hipMemcpy(c->d_team_deg16, deg16.data(), deg16.size() * 2, hipMemcpyHostToDevice));
    c->team_T = T; c->team_R = R; c->team_cap = at;
    return FORA_OK;
}

int team_fits(fora_ctx *c);
int ensure_workspace(fora_ctx *c, int want_slots, double omega_hint) {
    if (!c->n) return fail(c, FORA_E_ARG, "set_graph first");
    if (int rt = ensure_team(c)) return rt;
    if (!c->is_twin && c->opt_.hubs < 0 && !want_wide(c) && c->hub_for_team != (want_team(c) && c->team_T != 0)) {
        // the `team` / `team_size` options changed which push this graph takes: the hub copy follows (see build_hub_copy)
        std::vector<int32_t> col((size_t)std::max<int64_t>(1, c->nnz));
        HIPCHK(c, hipMemcpy(col.data(), c->d_col, (size_t)c->nnz * 4, hipMemcpyDeviceToHost));
        free_workspace(c);
        if (int rh = build_hub_copy(c, c->h_row_ptr.data(), col.data())) return rh;
        if (int rq = build_quad_copies(c)) return rq;
    }
    WsPlan p = plan_workspace(c, omega_hint, 1024); // bytes per slot hardly depend on the slot count (sub-bucket rounding)
    const uint64_t n = (uint64_t)c->n;
    // an existing workspace with the same layout is kept if it has enough slots: as many as the call can use, or as
    // many as an automatic plan would get at most (1024)
    auto keepable = [&](int need) {
        if (c->B <= 0 || c->B < need) return false;
        const WsPlan pe = plan_workspace(c, omega_hint, c->B);
        if ((c->team_T != 0) != (c->d_team_msg != nullptr)) return false;
        return c->binned == pe.binned && c->pbins == pe.pbins && c->seg_cap * sizeof(PushSeg) >= (uint64_t)c->B * pe.scratch &&
               c->wit_cap >= pe.wits && c->bk_cap == pe.bk_cap && c->sub == pe.sub;
    };
    {
        // slots the call can use: its queries, at most 1024, at most what memory allowed when the workspace was planned
        int need = c->batch_req > 0 ? c->batch_req : std::min(want_slots > 0 ? want_slots : 1024, 1024);
        if (c->batch_req == 0 && c->B > 0 && c->B_memcap > 0) need = std::min(need, c->B_memcap);
        if (keepable(need)) { const WsPlan pe = plan_workspace(c, omega_hint, c->B); return ensure_row_split(c, pe.nbins, pe.pbins); }
    }
    int B = c->batch_req > 0 ? c->batch_req : 0;
    if (B == 0) {
        // the slot count follows from the FREE memory: give the old workspace back first, or a re-plan (larger walk budget
        // of a top-k call, other sub-bucket count, more queries) would size itself from what the old one left over
        free_workspace(c);
        size_t fr = 0, tot = 0;
        HIPCHK(c, hipMemGetInfo(&fr, &tot));
        uint64_t budget = (uint64_t)(fr * (c->opt_.pipeline == 1 ? 0.4 : 0.75)); // with option pipeline a second lane holds its own workspace
        if (c->team_T) { // the team push's own buffers (allocated below) come out of the same memory
            const uint64_t T = c->team_T, nt = std::max<uint64_t>(1, (uint64_t)c->prop.multiProcessorCount * TEAM_WGS_PER_CU / T);
            const uint64_t team_bytes = nt * (2 * c->team_cap * 4 + 3 * T * (c->team_R + 64 + c->team_H) * 8 + T * (10ull << 17) + 2 * T * T * 8);
            budget -= std::min<uint64_t>(budget / 2, team_bytes);
        }
        B = (int)std::min<uint64_t>(1024, std::max<uint64_t>(1, budget / p.per_slot)); // ws, 1000 queries: 2845 q/s at 256, 3035 at 512, 3101 at 1000
    }
    B = std::max(1, B);
    const int memcap = B;
    if (want_slots > 0 && c->batch_req == 0) B = std::min(B, std::max(want_slots, 1));
    p = plan_workspace(c, omega_hint, B);
    const uint64_t scratch = (uint64_t)B * p.scratch;
    free_workspace(c);
    const uint64_t slab = (uint64_t)B * n;
    HIPCHK(c, hipMalloc(&c->d_residue, slab * 8));
    HIPCHK(c, hipMalloc(&c->d_ppr, slab * 8));
    if (p.binned) {
        HIPCHK(c, hipMalloc(&c->d_fl[0], slab * 4));
        HIPCHK(c, hipMalloc(&c->d_fl[1], slab * 4));
        HIPCHK(c, hipMalloc(&c->d_fl_count, (size_t)B * 2 * 4 * CSTRIDE));
        HIPCHK(c, hipMalloc(&c->d_inc_tab[0], (uint64_t)B * p.segq_cap * 8));
        HIPCHK(c, hipMalloc(&c->d_inc_tab[1], (uint64_t)B * p.segq_cap * 8));
        c->ov_cap = (uint32_t)std::max<uint64_t>(262144, n / 8); // bucket-overflow list, scales with the graph
        if (c->opt_.ovcap > 0) c->ov_cap = (uint32_t)c->opt_.ovcap; // tests
        HIPCHK(c, hipMalloc(&c->d_ov_w, (uint64_t)B * c->ov_cap * 4));
        HIPCHK(c, hipMalloc(&c->d_ov_inc, (uint64_t)B * c->ov_cap * 8));
        HIPCHK(c, hipMalloc(&c->d_ov_count, 2 * (size_t)B * 4 * CSTRIDE));
        HIPCHK(c, hipMalloc(&c->d_ov_bin, 2 * (size_t)B * p.nbins * 4));
        if (!want_wide(c)) HIPCHK(c, hipMalloc(&c->d_bk_w, (uint64_t)B * p.pbins * p.sub * p.bk_cap * 4)); // wide: one 64-bit word per message in bk_inc
        HIPCHK(c, hipMalloc(&c->d_bk_inc, (uint64_t)B * p.pbins * p.sub * p.bk_cap * 8));
        HIPCHK(c, hipMalloc(&c->d_bk_count, (size_t)B * p.pbins * p.sub * 4));
        HIPCHK(c, hipHostMalloc(&c->h_flc, (size_t)FLC_RING * B * 4 * CSTRIDE));
        c->dbm_words = (uint32_t)((uint64_t)p.nbins << (bin_shift(c) - 6));
        HIPCHK(c, hipMalloc(&c->d_dbm, 2 * (size_t)B * c->dbm_words * 8));
        HIPCHK(c, hipMalloc(&c->d_dflag, 2 * (size_t)B * p.nbins * 4));
        if (c->opt_.defer > 0) HIPCHK(c, hipMalloc(&c->d_dl, 2 * slab * 4)); // k_push_tail's deferred lists: only with the option (changing it re-plans the workspace)
        if (c->hubs && c->hub_shift == bin_shift(c)) HIPCHK(c, hipMalloc(&c->d_hubsum, (size_t)B * p.sub * c->hubs * 8));
        if (c->team_T) { // team push: message buffers of every team (two parities), bucket counts, control words
            const uint32_t T = c->team_T;
            uint32_t nteams = std::max<uint32_t>(1, (uint32_t)c->prop.multiProcessorCount * TEAM_WGS_PER_CU / T);
            if (c->opt_.team_max > 0) nteams = std::min<uint32_t>(nteams, (uint32_t)c->opt_.team_max);
            size_t fr = 0, tot = 0;
            HIPCHK(c, hipMemGetInfo(&fr, &tot));
            // pops of one member in one slot (ws-sized graph at eps 0.5: 43 k on average); beyond it: rsvl.  Tight memory: shorter logs
            c->team_rlog_cap = 1u << 17;
            auto per_team_bytes = [&](uint32_t logcap) { // message buffers + increment tables (two parities), rsvl, reserve logs, words
                return 2 * c->team_cap * 4 + 3 * (uint64_t)T * (c->team_R + 64 + c->team_H) * 8 + (uint64_t)T * logcap * 10 + 2 * (uint64_t)T * T * 8;
            };
            while (c->team_rlog_cap > 1024 && per_team_bytes(c->team_rlog_cap) > fr / 4) c->team_rlog_cap /= 2;
            const uint64_t per_team = per_team_bytes(c->team_rlog_cap);
            nteams = (uint32_t)std::max<uint64_t>(1, std::min<uint64_t>(nteams, (uint64_t)(fr / 2) / std::max<uint64_t>(1, per_team)));
            HIPCHK(c, hipMalloc(&c->d_team_msg, (size_t)nteams * 2 * c->team_cap * 4 + 64 + diag::TEAM_MSG_PROBE_WORDS * 4)); // (probe words: 0 in the product build)
            HIPCHK(c, hipMalloc(&c->d_team_inct, (size_t)nteams * 2 * T * (c->team_R + 64 + c->team_H) * 8));
            HIPCHK(c, hipMalloc(&c->d_team_rsvl, (size_t)nteams * T * c->team_R * 8));
            HIPCHK(c, hipMemset(c->d_team_rsvl, 0, (size_t)nteams * T * c->team_R * 8)); // every slot leaves it zero again
            HIPCHK(c, hipMalloc(&c->d_team_rlog_id, (size_t)nteams * T * c->team_rlog_cap * 2));
            HIPCHK(c, hipMalloc(&c->d_team_rlog_val, (size_t)nteams * T * c->team_rlog_cap * 8));
            HIPCHK(c, hipMalloc(&c->d_team_cnt, (size_t)nteams * 2 * T * T * 8));
            HIPCHK(c, hipMalloc(&c->d_team_ctl, (64 + (size_t)nteams * 5 * 16 * 2 + (size_t)nteams * ((size_t)B + 2)) * 4));
            c->team_n = nteams;
            if (int rf = team_fits(c)) return rf;
        }
    } else {
        HIPCHK(c, hipMalloc(&c->d_wl[0], slab * 8));
        HIPCHK(c, hipMalloc(&c->d_wl[1], slab * 8));
    }
    HIPCHK(c, hipMalloc(&c->d_scratch, scratch));
    HIPCHK(c, hipMalloc(&c->d_wit_count, (size_t)B * 4 * CSTRIDE));
    HIPCHK(c, hipMalloc(&c->d_sw, 2 * (size_t)B * 4 * CSTRIDE));
    HIPCHK(c, hipMemset(c->d_sw, 0, 2 * (size_t)B * 4 * CSTRIDE)); // self-resetting
    HIPCHK(c, hipMalloc(&c->d_tile_ctr, 2 * (size_t)B * 4 * CSTRIDE));
    HIPCHK(c, hipMemset(c->d_tile_ctr, 0, 2 * (size_t)B * 4 * CSTRIDE)); // every launch zeroes the set of the next one
    HIPCHK(c, hipMalloc(&c->d_counters, N_COUNTERS * sizeof(unsigned long long)));
    HIPCHK(c, hipMalloc(&c->d_qs, (size_t)B * sizeof(QState)));
    HIPCHK(c, hipMalloc(&c->d_src, (size_t)B * sizeof(int32_t)));
    HIPCHK(c, hipMalloc(&c->d_err, sizeof(uint32_t)));
    HIPCHK(c, hipHostMalloc(&c->h_pinned, (MAX_LEVELS + 2) * sizeof(unsigned long long)));
    HIPCHK(c, hipHostMalloc(&c->h_qs_pin, (size_t)B * sizeof(QState)));
    HIPCHK(c, hipHostMalloc(&c->h_steps_pin, sizeof(unsigned long long)));
    c->B = B;
    c->B_memcap = memcap;
    c->binned = p.binned; c->nbins = p.nbins; c->pbins = p.pbins; c->bk_cap = p.bk_cap; c->sub = p.sub; c->segq_cap = p.segq_cap;
    if (int rs = ensure_row_split(c, p.nbins, p.pbins)) return rs;
    c->wl_cap = slab;
    c->seg_cap = scratch / sizeof(PushSeg);
    c->wit_cap = p.wits; // per slot
    c->h_qs.resize(B);
    return FORA_OK;
}

Dev make_dev(fora_ctx *c, int nq, bool with_idx, double rmax = -1, double omega = -1) {
    if (rmax < 0) rmax = c->rmax;
    if (omega < 0) omega = c->omega;
    Dev d{};
    d.n = c->n; d.nq = nq;
    d.rowinfo = c->d_rowinfo; d.row_ptr = c->d_row_ptr; d.col = c->d_col; d.deg = c->d_deg;
    d.rp32 = c->d_rp32; d.colp = c->d_colp; d.colbits = c->colbits;
    d.colp32 = (uint64_t)c->nnz * c->colbits < (1ull << 32) ? 1 : 0;
    d.dg = c->dg;
    d.residue = c->d_residue; d.ppr = c->d_ppr;
    d.wl[0] = c->d_wl[0]; d.wl[1] = c->d_wl[1]; d.wl_cap = c->wl_cap;
    d.seg = (PushSeg *)c->d_scratch; d.seg_cap = c->seg_cap;
    d.wit = (WalkItemP *)c->d_scratch; d.wit_cap = c->wit_cap;
    d.wl_count = c->d_counters;
    d.seg_count = c->d_counters + (MAX_LEVELS + 2);
    d.wit_count = c->d_wit_count;
    d.tot_steps = c->d_counters + 2 * (size_t)(MAX_LEVELS + 2) + 1;
    d.qs = c->d_qs; d.src = c->d_src; d.err = c->d_err;
    d.afix = (uint64_t)std::ldexp(c->alpha, 62);
    double t = std::ceil(std::ldexp(rmax, 62));
    d.t1 = t >= 9223372036854775808.0 ? (~0ull >> 1) : (t < 1.0 ? 1 : (uint64_t)t);
    d.alpha32 = (uint32_t)(c->alpha * 4294967296.0);
    d.seed_lo = (uint32_t)c->seed; d.seed_hi = (uint32_t)(c->seed >> 32);
    d.alpha = c->alpha; d.omega = omega; d.opt = c->opt;
    d.binned = c->binned ? 1 : 0; d.nbins = c->nbins; d.wide = c->binned && want_wide(c) ? 1 : 0;
    d.pbins = c->pbins; d.bin_lo = 0; d.bin_cnt = std::min(c->pbins, c->nbins);
    d.col_push = c->d_col_push ? c->d_col_push : c->d_col;
    d.row_split = c->d_row_split;
    d.npass = c->pbins > 0 ? (c->nbins + c->pbins - 1) / c->pbins : 1;
    d.pass = 0;
    d.acc_group = 1; // (set per launch: acc_grid)
    // Dispatch order of the wide kernels (Dev::slot_major), measured per kernel (profiles/r06_slot_major.txt): slot-major pays for the bin kernel
    // when a launch holds many slots (LJ-sized, 143 slots: 1135-1205 -> 1073-1107 ms per 1000 queries in A/B runs; Twitter-2010-sized, 8 slots:
    // 620 -> 930 ms) and for the indexed walks of query calls (LJ-sized ~ - 5 %, Twitter-2010-sized - 2 %); it loses for the accumulate, the walk
    // allocation and everything in the top-k drivers.  FETCH_SIZE and TCC hits / misses are the same in both orders: if it is reuse, it is in the memory-side Infinity Cache.
    d.slot_major = c->opt_.slot_major >= 0 ? (uint32_t)c->opt_.slot_major & 15u
                   : (c->bk_div > 1 ? 0u : (4u | (nq >= 32 ? 1u : 0u)));
    d.tiny_max = (uint32_t)std::min<int64_t>(std::max<int64_t>(c->opt_.tiny, 0), 1023); // 512: ws accum 116 -> 113 ms per 3000 queries against 128; 2048: 119, 8192: 193 (the crossing list of the small-bucket path holds 1024)
    d.fl[0] = c->d_fl[0]; d.fl[1] = c->d_fl[1];
    d.fl_count[0] = c->d_fl_count; d.fl_count[1] = c->d_fl_count ? c->d_fl_count + (size_t)c->B * CSTRIDE : nullptr;
    d.inc_tab[0] = c->d_inc_tab[0]; d.inc_tab[1] = c->d_inc_tab[1]; d.segq_cap = c->segq_cap;
    d.pop_next = 1;
    d.stamps = c->d_stamps;
    d.round_div = 0;
    d.rounds = 1; // the query / push entry points raise it (k_round_sweep); top-k, --balanced and power iteration drive their own rounds
    if (c->binned && c->d_col_hub && c->d_hubsum && c->hub_shift == bin_shift(c) && c->pbins >= c->nbins) { // one pass per level only: the passes of larger graphs read a row-sorted copy
        d.col_hub = c->d_col_hub; d.hub_node = c->d_hub_node; d.hub_first = c->d_hub_first; d.hubsum = c->d_hubsum; d.hubs = c->hubs;
        d.hub_min = (uint32_t)std::min<int64_t>(std::max<int64_t>(c->opt_.hub_min, 1), 0x7FFFFFFF);
        d.tail_hubs = c->opt_.tail_hubs != 0 && (size_t)c->hubs * 8 <= 40960 ? 1u : 0u; // (k_push_tail: 20 KiB of static LDS + the sums within 64 KiB)
    }
    if (d.wide && c->d_col4 && !c->d_row_split && !c->d_col_push && c->opt_.quads != 0) { // one bin pass per level: the bin kernel reads quads
        d.col4 = c->d_col4; d.rowinfo4 = c->d_rowinfo4;
        d.col_hub4 = d.col_hub ? c->d_col_hub4 : nullptr;
        if (d.col_hub && !d.col_hub4) d.col4 = nullptr; // (no padded hub copy: edges one by one)
    }
    d.defer_k = TEST_PATHS && c->binned && c->d_dl ? (int32_t)std::min<int64_t>(std::max<int64_t>(c->opt_.defer, 0), 8) : 0; // the direct path keeps plain levels
    d.defer_min = (uint32_t)std::min<int64_t>(std::max<int64_t>(c->opt_.defer_min, 0), 0x7FFFFFFF);
    d.dbm[0] = c->d_dbm; d.dbm[1] = c->d_dbm ? c->d_dbm + (size_t)c->B * c->dbm_words : nullptr; d.dbm_words = c->dbm_words;
    d.dflag[0] = c->d_dflag; d.dflag[1] = c->d_dflag ? c->d_dflag + (size_t)c->B * c->nbins : nullptr;
    d.dl[0] = c->d_dl; d.dl[1] = c->d_dl ? c->d_dl + (size_t)c->B * c->n : nullptr;
    d.sw_count = c->d_sw; d.sw_done = c->d_sw ? c->d_sw + (size_t)c->B * CSTRIDE : nullptr;
    d.tile_ctr[0] = c->d_tile_ctr; d.tile_ctr[1] = c->d_tile_ctr ? c->d_tile_ctr + (size_t)c->B * CSTRIDE : nullptr;
    d.ov_w = c->d_ov_w; d.ov_inc = c->d_ov_inc; d.ov_cap = c->ov_cap;
    d.ov_count[0] = c->d_ov_count; d.ov_count[1] = c->d_ov_count ? c->d_ov_count + (size_t)c->B * CSTRIDE : nullptr;
    d.ov_bin[0] = c->d_ov_bin; d.ov_bin[1] = c->d_ov_bin ? c->d_ov_bin + (size_t)c->B * c->nbins : nullptr;
    d.bk_w = c->d_bk_w; d.bk_inc = c->d_bk_inc; d.bk_count = c->d_bk_count; d.bk_cap = c->bk_cap; d.sub = c->sub;
    if (with_idx) { d.rw_idx = c->d_rw_idx; d.idx_off = c->d_idx_off; d.idx_cnt = c->d_idx_cnt; }
    return d;
}

// ---- event-pair timing of individual launches on the ctx stream
int ev_begin(fora_ctx *c, int kind) {
    if (!c->profiling) return -1;
    if (c->ev_used == c->ev_pool.size()) {
        EvPair p{};
        if (hipEventCreate(&p.a) != hipSuccess || hipEventCreate(&p.b) != hipSuccess) return -1;
        c->ev_pool.push_back(p);
    }
    EvPair &p = c->ev_pool[c->ev_used];
    p.kind = kind;
    (void)hipEventRecord(p.a, c->stream);
    return (int)c->ev_used++;
}
void ev_end(fora_ctx *c, int h) {
    if (h >= 0) (void)hipEventRecord(c->ev_pool[h].b, c->stream);
}
void ev_collect(fora_ctx *c) { // call after the stream is idle
    for (size_t i = 0; i < c->ev_used; i++) {
        float ms = 0;
        if (hipEventElapsedTime(&ms, c->ev_pool[i].a, c->ev_pool[i].b) != hipSuccess) continue;
        switch (c->ev_pool[i].kind) {
        case 0: c->timing.push_pop_ms += ms; c->timing.push_pop_launches++; break;
        case 1: c->timing.push_expand_ms += ms; c->timing.push_expand_launches++; break;
        case 2: c->timing.walk_alloc_ms += ms; break;
        case 3: c->timing.walk_ms += ms; c->timing.walk_launches++; break;
        case 4: c->timing.other_ms += ms; break;
        case 5: c->timing.batch_ms += ms; c->timing.batches++; break;
        case 6: c->timing.push_accum_ms += ms; c->timing.push_accum_launches++; break;
        case 7: c->timing.walk_accum_ms += ms; break;
        case 9: c->timing.push_tail_ms += ms; c->timing.push_tail_launches++; break;
        case 10: c->timing.push_team_ms += ms; c->timing.push_team_launches++; break;
        case 8: c->timing.push_accum_ms += ms; break; // k_round_sweep: part of the level's accumulate time, not a launch of its own in the counts
        }
    }
    c->ev_used = 0;
}

int check_dev_err(fora_ctx *c) {
    uint32_t e = 0;
    HIPCHK(c, hipMemcpyAsync(&e, c->d_err, sizeof(e), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    c->bucket_overflow = (e & ERR_BUCKET_OVERFLOW) != 0;
    c->team_timeout_seen = (e & ERR_TEAM_TIMEOUT) != 0;
    if (e) c->team_dirty = true;
    if (e) {
        char buf[256];
        snprintf(buf, sizeof(buf), "device work list overflow (flags 0x%x%s)", e,
                 (e & ERR_TEAM_TIMEOUT) ? ": a team of k_push_team waited too long for a member (workgroups not co-resident); the call is run again with the bucketed push"
                 : (e & ERR_BUCKET_OVERFLOW) ? ": message buckets and their overflow list are full, raise FORA_HIP_BKCAP" : "");
        return fail(c, FORA_E_OVERFLOW, buf);
    }
    return FORA_OK;
}

// Level loop of the push for the slots already initialised (level-0 frontier in place).
// Launches run ahead of the host by SPEC levels: an empty level costs a few near-empty
// launches, a host round trip per level would cost more.
static inline size_t tail_lds(const Dev &d) { return d.tail_hubs && d.col_hub ? (size_t)d.hubs * 8 : 0; } // k_push_tail's dynamic LDS

// Wide accumulate: bins per workgroup (Dev::acc_group) and the grid's x size.  One bin per workgroup for query / power-iteration
// calls; the top-k drivers on wide graphs (many slots, rounds that touch a handful of bins: 94 k workgroups per launch at
// Twitter-2010 size, nearly all of them empty) take up to 8 consecutive bins of a slot per workgroup -- k_accum<false> 58 -> 39 ms
// per 125-source step, 301 -> 318 q/s.  (For LJ-sized queries, same 94 k workgroups but mostly busy ones: 211.7 ms grouped against
// 206.6 -- the bins of a group run one after the other --, hence not there.)
static unsigned acc_grid(const fora_ctx *c, Dev &dp, int nq) {
    const uint64_t pairs = (uint64_t)dp.bin_cnt * (uint64_t)std::max(1, nq);
    const uint64_t want = (uint64_t)std::max(1, c->prop.multiProcessorCount) * 24; // workgroups that keep the chip busy with one per CU at a time
    uint32_t g = c->bk_div > 1 ? (uint32_t)std::min<uint64_t>(std::max<uint64_t>(pairs / want, 1), 8) : 1u;
    if (c->opt_.acc_group > 0) g = (uint32_t)std::min<int64_t>(c->opt_.acc_group, ACC_GROUP_MAX);
    dp.acc_group = g;
    return (unsigned)((dp.bin_cnt + g - 1) / g);
}
int run_push_levels(fora_ctx *c, const Dev &d, uint64_t *levels_run = nullptr, int level_cap = 0, bool round_start = false) {
    const int nq = d.nq;
    if (round_start && c->binned && level_cap <= 0 && d.rounds <= 1 && c->opt_.tail != 0) {
        // A round of the top-k / --balanced drivers starts from every node at or over the round's threshold
        // (k_topk_frontier), often a handful: when no slot's frontier is larger than what k_push_tail takes over at anyway,
        // the whole round runs inside that kernel -- one launch instead of two per level plus the look-ahead levels
        // (Twitter-2010-sized top-k: 391 level launches of mostly empty workgroups per 125 queries).
        const int64_t tail_auto = std::min<int64_t>(32768, std::max<int64_t>(2048, (int64_t)nq * 32));
        const uint32_t tail_max = (uint32_t)(c->opt_.tail < 0 ? tail_auto : c->opt_.tail);
        uint32_t *cnt = c->h_flc;
        HIPCHK(c, hipMemcpyAsync(cnt, d.fl_count[0], (size_t)nq * 4 * CSTRIDE, hipMemcpyDeviceToHost, c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));
        uint32_t fmax = 0;
        for (int i = 0; i < nq; i++) fmax = std::max(fmax, cnt[(size_t)i * CSTRIDE]);
        if (fmax == 0) { if (levels_run) *levels_run = 0; return FORA_OK; }
        if (fmax <= tail_max) {
            int h = ev_begin(c, 9);
            hipLaunchKernelGGL(k_push_tail, dim3(nq), dim3(TAIL_THREADS), tail_lds(d), c->stream, d, 0, 0);
            ev_end(c, h);
            c->timing.levels++;
            if (levels_run) *levels_run = 1;
            hipError_t e = hipStreamSynchronize(c->stream);
            if (e != hipSuccess) return fail(c, FORA_E_HIP, std::string("push: ") + hipGetErrorString(e));
            e = hipGetLastError(); // (the launch check of the level loop's epilogue)
            if (e != hipSuccess) return fail(c, FORA_E_HIP, std::string("push launch: ") + hipGetErrorString(e));
            return FORA_OK;
        }
    }
    hipEvent_t done[SPEC + 1];
    for (auto &e : done) HIPCHK(c, hipEventCreateWithFlags(&e, hipEventDisableTiming));
    int rc = FORA_OK;
    int L = 0;
    const unsigned xb = c->binned ? c->sub : 1u; // producer workgroups per slot = sub-buckets per bucket (Dev::bk_w); ws at 1000 slots: 4 -> 196 ms, 8 -> 178, 16 -> 163, 32 -> 174
    // frontier size (largest slot) from which k_push_tail takes over; 0: never.  ws, push of 1000 queries (round 2's
    // tail kernel: no agent-scope fences, 4 relaxations in flight per lane): 1024: 86.3 ms, 4096: 85.2, 16384: 82.5,
    // 32768: 81.9, 131072: 163 (one workgroup per slot cannot feed the peak levels)
    // The tail runs one workgroup per slot, so it only pays while the slots alone fill the chip: with the 14 slots of a
    // Twitter-2010-sized batch 32768 -> 2048 takes the tail from 128 ms to 11 ms per 28 queries (15.05 -> 15.81 q/s);
    // LJ-sized, 140 slots: 32768 -> 4096 takes it from 16.8 to 2.2 ms per 280 queries (the bucketed levels it
    // replaces cost less).  Default: 32 x the slot count, between 2048 and 32768.
    const int64_t tail_auto = std::min<int64_t>(32768, std::max<int64_t>(2048, (int64_t)nq * 32));
    const uint32_t tail_max = (uint32_t)(c->opt_.tail < 0 ? tail_auto : c->opt_.tail);
    bool past_peak = c->opt_.tail_always == 1; // tests: do not wait for the frontier to have been large first
    for (;; L++) {
        if (level_cap > 0 && L >= level_cap) break; // power iteration: a fixed number of levels
        if (L >= MAX_LEVELS) { rc = fail(c, FORA_E_OVERFLOW, "push level cap reached"); break; }
        if (c->binned) {
            for (int lo = 0; lo < c->nbins; lo += c->pbins) { // one pass per group of pbins bins (usually one)
                Dev dp = d;
                if (level_cap > 0 || d.rounds > 1) dp.defer_k = 0; // capped runs (power iteration) and threshold rounds keep plain levels
                dp.bin_lo = lo;
                dp.bin_cnt = std::min(c->pbins, c->nbins - lo);
                dp.pass = lo / c->pbins;
                dp.pop_next = !(level_cap > 0 && L + 1 >= level_cap); // a capped run leaves the last crossing nodes unpopped
                dp.launch_par = (int32_t)(c->bin_launches++ & 1);
                int h = ev_begin(c, 1);
                {
                    const size_t hub_lds = dp.col_hub ? (size_t)dp.hubs * 8 : 0;
                    const bool hub = dp.col_hub != nullptr, split = dp.row_split != nullptr, sched = TEST_PATHS && (dp.rounds > 1 || dp.defer_k > 0);
                    const dim3 bgrid = (dp.wide && (dp.slot_major & 1u)) ? dim3(nq, xb) : dim3(xb, nq); // (Dev::slot_major: the slot is the fastest-varying index)
#define FORA_BIN_LAUNCH(NBV, NT, HUBV, SPLITV, SCHEDV) hipLaunchKernelGGL((k_pushq_bin<NBV, HUBV, SPLITV, SCHEDV>), bgrid, dim3(NT), hub_lds, c->stream, dp, L)
#if FORA_TEST_PATHS
#define FORA_BIN_SCHED(NBV, NT) FORA_BIN_LAUNCH(NBV, NT, true, true, true)
#else
#define FORA_BIN_SCHED(NBV, NT) (void)0
#endif
#define FORA_BIN_QUAD(NBV, NT, HUBV) hipLaunchKernelGGL((k_pushq_bin<NBV, HUBV, false, false, true>), bgrid, dim3(NT), hub_lds, c->stream, dp, L)
#define FORA_BIN_PICK(NBV, NT) do { \
                    if (!sched && !split && dp.col4 && NBV > MAX_BINS) { if (hub) FORA_BIN_QUAD(NBV, NT, true); else FORA_BIN_QUAD(NBV, NT, false); break; } \
                    if (sched) FORA_BIN_SCHED(NBV, NT); /* schedule experiments: the everything instantiation (test library only) */ \
                    else if (hub && split) FORA_BIN_LAUNCH(NBV, NT, true, true, false); \
                    else if (hub) FORA_BIN_LAUNCH(NBV, NT, true, false, false); \
                    else if (split) FORA_BIN_LAUNCH(NBV, NT, false, true, false); \
                    else FORA_BIN_LAUNCH(NBV, NT, false, false, false); } while (0)
                    if (d.wide && c->pbins > MAX_BINS_WIDE) FORA_BIN_PICK(MAX_BINS_HUGE, BIN_THREADS_HUGE);
                    else if (d.wide) FORA_BIN_PICK(MAX_BINS_WIDE, BIN_THREADS_WIDE);
                    else FORA_BIN_PICK(MAX_BINS, BLOCK);
#undef FORA_BIN_PICK
#undef FORA_BIN_QUAD
#undef FORA_BIN_SCHED
#undef FORA_BIN_LAUNCH
                }
                ev_end(c, h);
                h = ev_begin(c, 6);
                if (d.wide) { const unsigned gx = acc_grid(c, dp, nq); hipLaunchKernelGGL((k_accum<false, true>), (dp.slot_major & 2u) ? dim3(nq, gx) : dim3(gx, nq), dim3(ACC_THREADS_WIDE), 0, c->stream, dp, L); }
                else hipLaunchKernelGGL((k_accum<false, false>), dim3(dp.bin_cnt, nq), dim3(ACC_THREADS), 0, c->stream, dp, L);
                ev_end(c, h);
            }
            if (TEST_PATHS && d.rounds > 1) { // threshold rounds: slots whose frontier ran dry move on to the next (halved) threshold
#if FORA_TEST_PATHS
                int h = ev_begin(c, 8);
                hipLaunchKernelGGL(k_round_sweep, dim3(std::min<uint32_t>(slab_grid_x(c, nq), 32u), nq), dim3(BLOCK), 0, c->stream, d, L);
                ev_end(c, h);
#endif
            }
            (void)hipMemcpyAsync(c->h_flc + (size_t)((L + 1) % FLC_RING) * c->B * CSTRIDE, d.fl_count[(L + 1) & 1],
                                 (size_t)nq * 4 * CSTRIDE, hipMemcpyDeviceToHost, c->stream);
        } else {
            int h = ev_begin(c, 0);
            hipLaunchKernelGGL(k_push_pop, dim3(c->grid_blocks), dim3(BLOCK), 0, c->stream, d, L);
            ev_end(c, h);
            h = ev_begin(c, 1);
            hipLaunchKernelGGL(k_push_expand, dim3(c->grid_blocks), dim3(BLOCK), 0, c->stream, d, L);
            ev_end(c, h);
            (void)hipMemcpyAsync(&c->h_pinned[L + 1], &d.wl_count[L + 1], sizeof(unsigned long long),
                                 hipMemcpyDeviceToHost, c->stream);
        }
        c->timing.levels++;
        (void)hipEventRecord(done[L % (SPEC + 1)], c->stream);
        if (L >= SPEC) {
            const int K = L - SPEC;
            if (hipEventSynchronize(done[K % (SPEC + 1)]) != hipSuccess) { rc = fail(c, FORA_E_HIP, "event sync"); break; }
            bool empty;
            uint32_t fmax = 0;
            if (c->binned) {
                empty = true;
                const uint32_t *cnt = c->h_flc + (size_t)((K + 1) % FLC_RING) * c->B * CSTRIDE;
                uint32_t rounds_left = 0; // word 1 of a slot's counter line: threshold rounds still to come (k_round_sweep)
                for (int i = 0; i < nq; i++) { // word 2: nodes the level deferred (they are part of the work that is left)
                    fmax = std::max(fmax, cnt[(size_t)i * CSTRIDE] + cnt[(size_t)i * CSTRIDE + 2]);
                    rounds_left = std::max(rounds_left, cnt[(size_t)i * CSTRIDE + 1]);
                }
                empty = fmax == 0 && rounds_left == 0;
                if (rounds_left) fmax = std::max(fmax, tail_max + 1); // k_push_tail knows the final threshold only
            } else {
                empty = c->h_pinned[K + 1] == 0;
            }
            if (fmax > tail_max) past_peak = true; // the first levels are small too, but growing
            if (!empty && c->binned && tail_max > 0 && past_peak && fmax <= tail_max) {
                // every slot's frontier is small: finish inside one workgroup per slot instead of launching levels
                const int next = L + 1;
                const int remaining = level_cap > 0 ? level_cap - next : 0;
                if (level_cap <= 0 || remaining > 0) {
                    int h = ev_begin(c, 9);
                    Dev dt = d;
                    if (d.rounds > 1) dt.defer_k = 0;
                    hipLaunchKernelGGL(k_push_tail, dim3(nq), dim3(TAIL_THREADS), tail_lds(dt), c->stream, dt, next, remaining);
                    ev_end(c, h);
                    c->timing.levels++;
                }
                break;
            }
            if (empty) break;
        }
    }
    hipError_t e = hipStreamSynchronize(c->stream);
    for (auto &ev : done) (void)hipEventDestroy(ev);
    if (levels_run) *levels_run = (uint64_t)L + 1;
    if (rc == FORA_OK && e != hipSuccess) rc = fail(c, FORA_E_HIP, std::string("push: ") + hipGetErrorString(e));
    if (rc == FORA_OK) {
        e = hipGetLastError();
        if (e != hipSuccess) rc = fail(c, FORA_E_HIP, std::string("push launch: ") + hipGetErrorString(e));
    }
    return rc;
}

// Team push of a batch (fora_team.h): ONE launch runs every slot's push down to a small frontier with the residue
// resident in LDS, k_push_tail finishes the slots.  Nothing here waits for the device.
static bool use_team(const fora_ctx *c, const Dev &d) {
    // not while a second lane may have its own full-chip team kernel in flight (option pipeline), not after a time-out
    return c->team_T && c->d_team_msg && c->binned && !d.wide && !c->balanced && d.rounds <= 1 && d.defer_k == 0 && want_team(c) &&
           c->team_fit != 0 && c->team_suspend == 0 && c->opt_.pipeline != 1 && !c->is_twin;
}
// Can every workgroup of a k_push_team launch be resident at once?  (Asked once per workspace; raises the kernel's
// dynamic LDS limit on the way.)
int team_fits(fora_ctx *c) {
    if (c->team_fit >= 0 || !c->team_T || !c->d_team_msg) return FORA_OK;
    const size_t lds = ((size_t)c->team_R + 1 + c->team_H) * 8;
    hipFuncAttributes fa{};
    HIPCHK(c, hipFuncGetAttributes(&fa, (const void *)k_push_team));
    HIPCHK(c, hipFuncSetAttribute((const void *)k_push_team, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(163840 - (int)fa.sharedSizeBytes)));
    int per_cu = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, (const void *)k_push_team, TEAM_THREADS, lds) != hipSuccess) { (void)hipGetLastError(); per_cu = 0; }
    const uint32_t grid = c->team_n * c->team_T;
    c->team_fit = (uint64_t)per_cu * (uint64_t)c->prop.multiProcessorCount >= grid ? 1 : 0;
    int coop = 0;
    if (hipDeviceGetAttribute(&coop, hipDeviceAttributeCooperativeLaunch, c->device) != hipSuccess) { (void)hipGetLastError(); coop = 0; }
    c->team_coop_ok = c->opt_.team_coop == 1 && coop != 0;
    return FORA_OK;
}
int run_push_team(fora_ctx *c, const Dev &d) {
    const uint32_t T = c->team_T, nteams = c->team_n;
    TeamDev a{};
    a.n = d.n; a.nq = d.nq; a.rowinfo = d.rowinfo; a.row_ptr = d.row_ptr; a.deg = d.deg; a.src = d.src;
    a.residue = d.residue; a.ppr = d.ppr; a.fl0 = d.fl[0]; a.fl_count0 = d.fl_count[0]; a.inc_tab0 = d.inc_tab[0];
    a.segq_cap = d.segq_cap; a.qs = d.qs; a.err = d.err; a.afix = d.afix; a.t1 = d.t1;
    a.T = T; a.R = c->team_R; a.nteams = nteams;
    a.colt = c->d_colt; a.rowq = c->d_team_rowq; a.n2l = c->d_team_n2l; a.l2n = c->d_team_l2n; a.deg16 = c->d_team_deg16; a.rowl = c->d_team_rowl; a.rsvl = c->d_team_rsvl; a.rlog_id = c->d_team_rlog_id; a.rlog_val = c->d_team_rlog_val; a.rlog_cap = c->opt_.team_log == 0 ? 0u : c->opt_.team_log > 0 ? std::min<uint32_t>((uint32_t)c->opt_.team_log, c->team_rlog_cap) : c->team_rlog_cap; a.H = c->team_H; a.hubtgt = c->d_team_hubtgt; a.off = c->d_team_off; a.msg = c->d_team_msg; a.inct = c->d_team_inct; a.cntw = c->d_team_cnt;
    a.ctl = c->d_team_ctl;
    a.sync = (unsigned long long *)(c->d_team_ctl + 64);
    a.slot_seq = c->d_team_ctl + 64 + (size_t)nteams * 5 * 16 * 2;
    // frontier size of a slot at which k_push_tail (one workgroup per slot, global atomics) takes over; 0: never
    const int64_t tail_auto = 4096;
    a.tail_max = (uint32_t)std::min<int64_t>(std::max<int64_t>(c->opt_.team_tail < 0 ? tail_auto : c->opt_.team_tail, 0), 0x7FFFFFFF);
    if (c->opt_.tail == 0) a.tail_max = 0; // `tail` 0 keeps k_push_tail out of every path (tests)
    a.tail_always = c->opt_.tail_always == 1 ? 1u : 0u;
    const uint32_t grid = nteams * T;
    a.xcd = (c->opt_.team_xcd >= 1 && grid % 8 == 0 && (grid / 8) % T == 0) ? (uint32_t)c->opt_.team_xcd : 0u;
    a.stamps = c->d_stamps;
    a.abort_level = (uint32_t)std::min<int64_t>(std::max<int64_t>(c->opt_.team_abort_level, 0), 1 << 20);
    a.timeout_ticks = (uint64_t)std::min<int64_t>(std::max<int64_t>(c->opt_.team_timeout_ms, 0), 60000) * 100000ull; // (100 MHz wall clock)
    const size_t lds = ((size_t)a.R + 1 + a.H) * 8;
    if (c->team_dirty) { HIPCHK(c, hipMemsetAsync(c->d_team_rsvl, 0, (size_t)nteams * T * c->team_R * 8, c->stream)); c->team_dirty = false; }
    HIPCHK(c, hipMemsetAsync(c->d_team_ctl, 0, (64 + (size_t)nteams * 5 * 16 * 2) * 4, c->stream));
    HIPCHK(c, hipMemsetAsync(c->d_team_cnt, 0, (size_t)nteams * 2 * T * T * 8, c->stream)); // no barrier tag of an earlier launch
    HIPCHK(c, hipMemsetAsync(a.slot_seq, 0xFF, (size_t)nteams * ((size_t)d.nq + 2) * 4, c->stream));
    int h = ev_begin(c, 10);
    // The members of a team spin on each other: every workgroup of the launch must be resident at once.  team_fits() has
    // checked that the grid fits the device; a cooperative launch makes the runtime promise it (and keeps cooperative
    // kernels of other contexts from interleaving their workgroups with ours).  Whatever still goes wrong ends in
    // ERR_TEAM_TIMEOUT after team_timeout_ms, and with_retry runs the call again through the bucketed kernels.
    bool launched = false;
    if (c->team_coop_ok && !c->team_coop_failed) {
        void *args[] = {(void *)&a};
        const hipError_t le = hipLaunchCooperativeKernel((const void *)k_push_team, dim3(grid), dim3(TEAM_THREADS), args, (unsigned)lds, c->stream);
        if (le == hipSuccess) launched = true;
        else { (void)hipGetLastError(); c->team_coop_failed = true; }
    }
    if (!launched) hipLaunchKernelGGL(k_push_team, dim3(grid), dim3(TEAM_THREADS), lds, c->stream, a);
    ev_end(c, h);
    c->timing.levels++;
    if (a.tail_max) {
        h = ev_begin(c, 9);
        hipLaunchKernelGGL(k_push_tail, dim3(d.nq), dim3(TAIL_THREADS), tail_lds(d), c->stream, d, 0, 0);
        ev_end(c, h);
        c->timing.levels++;
    }
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(c, FORA_E_HIP, std::string("team push launch: ") + hipGetErrorString(e));
    return FORA_OK;
}

// k_topk_select over the slabs of `ds.ppr`; large graphs first compact each slot's non-zero entries (in id order, so
// ties keep resolving to the lowest ids) into the push's frontier / increment buffers, which are idle here.
constexpr unsigned NZ_X = 1024;
int launch_select(fora_ctx *c, const Dev &ds, int nb, int k, int32_t *ids, double *scores, int raw, const double *h_thr = nullptr) {
    bool compact = c->binned && c->n >= (1 << 20);
    if (c->opt_.select_compact >= 0) compact = c->binned && c->opt_.select_compact == 1; // tests: force / forbid the compacted form
    if (!compact) {
        hipLaunchKernelGGL(k_topk_select, dim3(nb), dim3(SEL_THREADS), 0, c->stream, ds, k, ids, scores, raw,
                           (const uint32_t *)nullptr, (const uint64_t *)nullptr, (const uint32_t *)nullptr);
        return FORA_OK;
    }
    if (!c->d_nz_counts) HIPCHK(c, hipMalloc(&c->d_nz_counts, (size_t)c->B * (NZ_X + 1) * 4));
    const unsigned X = (unsigned)std::min<uint64_t>(NZ_X, ((uint64_t)c->n + 4095) / 4096);
    const uint32_t R = (uint32_t)(((uint64_t)c->n + X - 1) / X);
    uint32_t *ccount = c->d_nz_counts + (size_t)c->B * NZ_X;
    const double *thr = nullptr; // per-slot lower limit of the entries worth compacting (see k_nz_count)
    if (h_thr && !raw) {
        if (!c->d_sel_thr) HIPCHK(c, hipMalloc(&c->d_sel_thr, (size_t)c->B * sizeof(double)));
        HIPCHK(c, hipMemcpyAsync(c->d_sel_thr, h_thr, (size_t)nb * sizeof(double), hipMemcpyHostToDevice, c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream)); // h_thr is the caller's pageable buffer
        thr = c->d_sel_thr;
    }
    hipLaunchKernelGGL(k_nz_count, dim3(X, nb), dim3(BLOCK), 0, c->stream, ds, R, c->d_nz_counts, thr);
    hipLaunchKernelGGL(k_nz_write, dim3(X, nb), dim3(BLOCK), 0, c->stream, ds, R, (const uint32_t *)c->d_nz_counts, c->d_fl[0],
                       c->d_inc_tab[0], ccount, thr);
    hipLaunchKernelGGL(k_topk_select, dim3(nb), dim3(SEL_THREADS), 0, c->stream, ds, k, ids, scores, raw,
                       (const uint32_t *)c->d_fl[0], (const uint64_t *)c->d_inc_tab[0], (const uint32_t *)ccount);
    return FORA_OK;
}

// per-level bookkeeping of the bucketed push that must start from zero
int reset_binned_counters(fora_ctx *c) {
    if (!c->binned) return FORA_OK;
    HIPCHK(c, hipMemsetAsync(c->d_fl_count, 0, (size_t)c->B * 2 * 4 * CSTRIDE, c->stream));
    HIPCHK(c, hipMemsetAsync(c->d_bk_count, 0, (size_t)c->B * c->pbins * c->sub * 4, c->stream));
    HIPCHK(c, hipMemsetAsync(c->d_ov_count, 0, 2 * (size_t)c->B * 4 * CSTRIDE, c->stream));
    HIPCHK(c, hipMemsetAsync(c->d_ov_bin, 0, 2 * (size_t)c->B * c->nbins * 4, c->stream));
    if (TEST_PATHS) { // bounded deferral's bitmaps and flags (test library only; 385 MB per top-k round at Twitter-2010 size)
        HIPCHK(c, hipMemsetAsync(c->d_dbm, 0, 2 * (size_t)c->B * c->dbm_words * 8, c->stream)); // (a complete push leaves them clear; an aborted one may not)
        HIPCHK(c, hipMemsetAsync(c->d_dflag, 0, 2 * (size_t)c->B * c->nbins * 4, c->stream));
    }
    HIPCHK(c, hipMemsetAsync(c->d_tile_ctr, 0, 2 * (size_t)c->B * 4 * CSTRIDE, c->stream)); // both parity sets, every slot: a launch only re-zeroes the slots it runs
    return FORA_OK;
}

int reset_batch_state(fora_ctx *c, int nq, const int32_t *sources) {
    const uint64_t bytes = (uint64_t)nq * c->n * 8;
    int h = ev_begin(c, 4);
    HIPCHK(c, hipMemsetAsync(c->d_residue, 0, bytes, c->stream));
    HIPCHK(c, hipMemsetAsync(c->d_ppr, 0, bytes, c->stream));
    HIPCHK(c, hipMemsetAsync(c->d_counters, 0, N_COUNTERS * sizeof(unsigned long long), c->stream));
    HIPCHK(c, hipMemsetAsync(c->d_err, 0, sizeof(uint32_t), c->stream));
    HIPCHK(c, hipMemsetAsync(c->d_wit_count, 0, (size_t)c->B * 4 * CSTRIDE, c->stream));
    HIPCHK(c, hipMemcpyAsync(c->d_src, sources, (size_t)nq * sizeof(int32_t), hipMemcpyHostToDevice, c->stream));
    int rc = reset_binned_counters(c);
    ev_end(c, h);
    return rc;
}

enum { RUN_PUSH_ONLY = 1 };

// slots per batch for nq queries on B slots: the fewest batches, all about the same size (1000 queries on 140 slots: 8 x 125,
// not 7 x 140 + 20 -- a small trailing batch costs almost a full batch's level latencies)
static int even_batch(int nq, int B) {
    if (nq <= B || B <= 0) return std::max(nq, 1);
    const int nbatch = (nq + B - 1) / B;
    return (nq + nbatch - 1) / nbatch;
}

// refinement launches after k_walk_alloc: indexed walks, online walks, and the accumulate of their results
void launch_walks(fora_ctx *c, const Dev &d, int nq, bool with_idx, uint32_t round, int nzh) {
    const dim3 wg(walk_grid_x(c, nq), nq);
    const dim3 wgs(c->binned ? c->sub : 1u, nq); // kernels that fill buckets: one workgroup per sub-bucket (Dev::bk_w)
    int h = ev_begin(c, 3);
    if (with_idx) {
        if (!c->binned) hipLaunchKernelGGL(k_walk_idx<1>, wg, dim3(BLOCK), 0, c->stream, d);
        else if (!d.wide) hipLaunchKernelGGL(k_walk_idx<MAX_BINS>, wgs, dim3(BLOCK), 0, c->stream, d);
        else
            for (int lo = 0; lo < c->nbins; lo += c->pbins) { // buckets are reused pass by pass
                Dev dp = d;
                dp.bin_lo = lo;
                dp.bin_cnt = std::min(c->pbins, c->nbins - lo);
                const dim3 wgi = (dp.slot_major & 4u) ? dim3(wgs.y, wgs.x) : wgs;
                if (c->pbins > MAX_BINS_WIDE) hipLaunchKernelGGL(k_walk_idx<MAX_BINS_HUGE>, wgi, dim3(BIN_THREADS_HUGE), 0, c->stream, dp);
                else hipLaunchKernelGGL(k_walk_idx<MAX_BINS_WIDE>, wgi, dim3(BIN_THREADS_WIDE), 0, c->stream, dp);
                { const unsigned gx = acc_grid(c, dp, nq); hipLaunchKernelGGL((k_accum<true, true>), (dp.slot_major & 2u) ? dim3(nq, gx) : dim3(gx, nq), dim3(ACC_THREADS_WIDE), 0, c->stream, dp, 0); }
            }
    }
    const bool dg = c->binned && !d.wide && d.dg.colp && c->opt_.walk_dg != 0; // narrow layout: one gather per step over the degree-grouped copy
    const bool xl = dg && c->opt_.walk_dg != 1 && d.dg.invb;                  // ... and results in bucket order (no gather per walk either)
    if (xl && with_idx) { // the indexed results are in plain ids: reduce them before the buckets are reused in bucket order
        ev_end(c, h);
        h = ev_begin(c, 7);
        hipLaunchKernelGGL((k_accum<true, false>), dim3(c->nbins, nq), dim3(ACC_THREADS), 0, c->stream, d, 0);
        ev_end(c, h);
        h = ev_begin(c, 3);
    }
    Dev dw = d;
    if (xl) { dw.nbins = (int32_t)d.dg.nbx; dw.acc_xl = d.dg.invb; }
    if (dg) {
        const size_t lds = (xl ? (size_t)((d.dg.H + 1) & ~1u) * 8 : 0) + (size_t)4 * d.dg.nrec * 4 + (((size_t)d.dg.nblk + 3) & ~(size_t)3); // hub sums | 16-byte records | block -> class bytes
#define FORA_DG_LAUNCH(NZH, B32, XLF) hipLaunchKernelGGL((k_walk_dg<NZH, B32, XLF>), wgs, dim3(DG_THREADS), lds, c->stream, dw, round)
        const int sel = (nzh ? 4 : 0) | (d.dg.bits32 ? 2 : 0) | (xl ? 1 : 0);
        switch (sel) {
        case 0: FORA_DG_LAUNCH(false, false, false); break;
        case 1: FORA_DG_LAUNCH(false, false, true); break;
        case 2: FORA_DG_LAUNCH(false, true, false); break;
        case 3: FORA_DG_LAUNCH(false, true, true); break;
        case 4: FORA_DG_LAUNCH(true, false, false); break;
        case 5: FORA_DG_LAUNCH(true, false, true); break;
        case 6: FORA_DG_LAUNCH(true, true, false); break;
        default: FORA_DG_LAUNCH(true, true, true); break;
        }
#undef FORA_DG_LAUNCH
    } else
        hipLaunchKernelGGL(k_walk_online<WALK_TO_PPR>, c->binned && !d.wide ? wgs : wg, dim3(BLOCK), 0, c->stream, d, round, nzh, (int32_t *)nullptr);
    ev_end(c, h);
    if (c->binned && !d.wide) { // narrow layout: indexed and online results share the buckets (bucket-order results: see above)
        h = ev_begin(c, 7);
        hipLaunchKernelGGL((k_accum<true, false>), dim3(dw.nbins, nq), dim3(ACC_THREADS), 0, c->stream, dw, 0);
        ev_end(c, h);
    }
}

// one batch of <= B sources: push (+ refinement).  Results stay in the slabs.
// one batch of <= B sources, part 1: push (host-driven level loop, returns when the push is done)
// and everything after it enqueued on the lane's stream.  Results stay in the slabs.
// --balanced push of a batch (query.h:848-884): rounds of the incremental push (algo.h:1020-1093) with rmax halving
// from 8*config.rmax; a slot keeps going while its estimated walk cost exceeds what its push has cost so far.
int push_balanced(fora_ctx *c, const int32_t *sources, int nq, bool with_idx) {
    if (!c->d_active) HIPCHK(c, hipMalloc(&c->d_active, (size_t)c->B));
    const uint32_t chunks = slab_grid_x(c, std::min(nq, c->B));
    Dev d = make_dev(c, nq, with_idx);
    int h = ev_begin(c, 4);
    hipLaunchKernelGGL(k_init_batch, dim3((nq + BLOCK - 1) / BLOCK), dim3(BLOCK), 0, c->stream, d, 1);
    ev_end(c, h);
    std::vector<uint8_t> active((size_t)nq, 1);
    std::vector<uint64_t> rsum_fix((size_t)nq, FIX_ONE), pops((size_t)nq, 0), relax((size_t)nq, 0);
    c->h_rmax_used.assign((size_t)nq, c->rmax);
    c->h_rounds.assign((size_t)nq, 1);
    for (int i = 0; i < nq; i++)
        if (c->h_row_ptr[sources[i] + 1] == c->h_row_ptr[sources[i]]) active[i] = 0; // :864, :882
    for (int i = 0; i < nq; i++) if (active[i]) c->h_rounds[i] = 0;
    double rmax = c->rmax * c->bal_start; // :862
    for (int round = 0;; round++) {
        bool any = false;
        for (int i = 0; i < nq; i++) {
            if (!active[i]) continue;
            const double t = (!with_idx || rmax >= c->rmax) ? c->t_walk : c->t_idx;                       // :825-838
            const double est = c->omega * std::ldexp((double)rsum_fix[i], -62) * (1 - c->alpha) * t;
            const double used = (double)pops[i] * c->c_pop + (double)relax[i] * c->c_edge;
            if (!(est > used)) active[i] = 0;                                                              // :866
            else { any = true; c->h_rmax_used[i] = rmax; c->h_rounds[i] = round + 1; }
        }
        if (!any) break;
        if (round >= 64) return fail(c, FORA_E_OVERFLOW, "--balanced: rmax halved 64 times");
        HIPCHK(c, hipMemcpyAsync(c->d_active, active.data(), (size_t)nq, hipMemcpyHostToDevice, c->stream));
        if (round) {
            HIPCHK(c, hipMemsetAsync(c->d_counters, 0, N_COUNTERS * sizeof(unsigned long long), c->stream));
            int rc = reset_binned_counters(c);
            if (rc) return rc;
        }
        Dev dr = make_dev(c, nq, with_idx, rmax, c->omega);
        h = ev_begin(c, 4);
        hipLaunchKernelGGL(k_topk_frontier, dim3(chunks, nq), dim3(BLOCK), 0, c->stream, dr, (const uint8_t *)c->d_active);
        ev_end(c, h);
        int rc = run_push_levels(c, dr, nullptr, 0, true);
        if (rc) return rc;
        HIPCHK(c, hipMemcpy(c->h_qs.data(), c->d_qs, (size_t)nq * sizeof(QState), hipMemcpyDeviceToHost));
        for (int i = 0; i < nq; i++) {
            rsum_fix[i] = FIX_ONE - c->h_qs[i].reserved;
            pops[i] = c->h_qs[i].pops;
            relax[i] = c->h_qs[i].relax;
        }
        rmax /= 2; // :875
    }
    return FORA_OK;
}

int batch_begin(fora_ctx *c, const int32_t *sources, int nq, bool with_idx, int flags) {
    for (int i = 0; i < nq; i++)
        if (sources[i] < 0 || sources[i] >= c->n) return fail(c, FORA_E_ARG, "source id out of range");
    const int hb = ev_begin(c, 5);
    int rc = reset_batch_state(c, nq, sources);
    if (rc) return rc;
    Dev d = make_dev(c, nq, with_idx);
    int h;
    if (c->balanced) {
        rc = push_balanced(c, sources, nq, with_idx);
    } else {
        if (TEST_PATHS && c->binned) d.rounds = (int32_t)std::min<int64_t>(std::max<int64_t>(c->opt_.rounds, 1), 16);
        d.round_div = (uint32_t)std::min<int64_t>(std::max<int64_t>(c->opt_.round_div, 0), 1 << 20);
        const bool team = use_team(c, d);
        h = ev_begin(c, 4);
        hipLaunchKernelGGL(k_init_batch, dim3((nq + BLOCK - 1) / BLOCK), dim3(BLOCK), 0, c->stream, d, team ? 3 : 0);
        ev_end(c, h);
        rc = team ? run_push_team(c, d) : run_push_levels(c, d);
        d.rounds = 1;
    }
    if (rc) return rc;
    if (!(flags & RUN_PUSH_ONLY)) {
        const uint32_t chunks = slab_grid_x(c, nq);
        h = ev_begin(c, 2);
        hipLaunchKernelGGL(k_walk_alloc<ALLOC_QUERY>, (d.wide && (d.slot_major & 8u)) ? dim3(nq, chunks) : dim3(chunks, nq), dim3(BLOCK), 0, c->stream, d, with_idx ? 1 : 0,
                           (const uint8_t *)nullptr, (uint64_t *)nullptr, (unsigned long long *)nullptr, 0u);
        ev_end(c, h);
        launch_walks(c, d, nq, with_idx, 0u, c->opt ? 1 : 0);
    }
    {
        const uint32_t chunks = (uint32_t)std::min<int64_t>(((int64_t)c->n + BLOCK - 1) / BLOCK, 64);
        h = ev_begin(c, 4);
        hipLaunchKernelGGL(k_ppr_sum, dim3(chunks, nq), dim3(BLOCK), 0, c->stream, d);
        ev_end(c, h);
    }
    HIPCHK(c, hipMemcpyAsync(c->h_qs_pin, c->d_qs, (size_t)nq * sizeof(QState), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipMemcpyAsync(c->h_steps_pin, d.tot_steps, sizeof(unsigned long long), hipMemcpyDeviceToHost, c->stream));
    ev_end(c, hb);
    c->pending_nq = nq;
    return FORA_OK;
}

// part 2: wait for the lane, check device flags, fold timings and counters
int batch_finish(fora_ctx *c) {
    const int nq = c->pending_nq;
    c->pending_nq = 0;
    int rc = check_dev_err(c);
    if (rc) return rc;
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(c, FORA_E_HIP, std::string("batch: ") + hipGetErrorString(e));
    ev_collect(c);
    for (int i = 0; i < nq; i++) c->h_qs[i] = c->h_qs_pin[i];
    c->timing.walk_steps += *c->h_steps_pin;
    for (int i = 0; i < nq; i++) {
        c->timing.pops += c->h_qs[i].pops;
        c->timing.relax += c->h_qs[i].relax;
        c->timing.walks += c->h_qs[i].n_walks;
        c->timing.idx_hits += c->h_qs[i].n_hit;
    }
    return FORA_OK;
}

void fill_stats(const fora_ctx *c, int nq, fora_query_stats *out) {
    for (int i = 0; i < nq; i++) {
        const QState &s = c->h_qs[i];
        fora_query_stats &o = out[i];
        o.rsum_fix = FIX_ONE - s.reserved;
        o.rsum = std::ldexp((double)o.rsum_fix, -62);
        o.n_rw = s.n_rw; o.n_walks = s.n_walks; o.n_idx_hit = s.n_hit;
        o.pops = s.pops; o.relax = s.relax; o.ppr_sum_fix = s.ppr_sum;
        o.levels = (int32_t)s.levels; o.dangling_source = (int32_t)s.dangling_source;
        o.rmax_used = c->balanced && (size_t)i < c->h_rmax_used.size() ? c->h_rmax_used[i] : c->rmax;
        o.push_rounds = c->balanced && (size_t)i < c->h_rounds.size() ? c->h_rounds[i] : 1;
        o.reserved_ = 0;
    }
}

// (re)creates the second lane and mirrors graph / index / params into it (non-owning pointers)
int sync_twin(fora_ctx *c) {
    if (!c->twin) {
        fora_ctx *w = new (std::nothrow) fora_ctx();
        if (!w) return fail(c, FORA_E_NOMEM, "twin lane");
        w->is_twin = true;
        w->device = c->device;
        w->prop = c->prop;
        if (hipStreamCreateWithFlags(&w->stream, hipStreamNonBlocking) != hipSuccess) { delete w; return fail(c, FORA_E_HIP, "twin stream"); }
        w->profiling = c->profiling;
        w->opt_ = c->opt_;
        w->grid_blocks = c->grid_blocks;
        c->twin = w;
    }
    fora_ctx *w = c->twin;
    if (w->n != c->n || w->d_col != c->d_col) free_workspace(w);
    w->n = c->n; w->m_attr = c->m_attr; w->nnz = c->nnz;
    w->d_row_ptr = c->d_row_ptr; w->d_col = c->d_col; w->d_rowinfo = c->d_rowinfo; w->d_deg = c->d_deg;
    w->d_rp32 = c->d_rp32; w->d_colp = c->d_colp; w->colbits = c->colbits;
    w->dg = c->dg; // arrays owned by c
    w->d_col_hub = c->d_col_hub; w->d_hub_node = c->d_hub_node; w->d_hub_first = c->d_hub_first; w->hubs = c->hubs; w->hub_shift = c->hub_shift;
    w->d_col_push = c->d_col_push; w->d_row_split = c->d_row_split; w->split_pbins = c->split_pbins; // shared, owned by c
    w->d_col4 = c->d_col4; w->d_col_hub4 = c->d_col_hub4; w->d_rowinfo4 = c->d_rowinfo4; w->quads = c->quads;
    w->d_colt = c->d_colt; w->d_team_rowq = c->d_team_rowq; w->d_team_off = c->d_team_off; w->d_team_n2l = c->d_team_n2l; w->d_team_l2n = c->d_team_l2n; w->d_team_deg16 = c->d_team_deg16; w->d_team_rowl = c->d_team_rowl; w->d_team_hubtgt = c->d_team_hubtgt; w->team_H = c->team_H; w->team_T = c->team_T; w->team_R = c->team_R; w->team_cap = c->team_cap; w->dangling_frac = c->dangling_frac;
    w->have_params = c->have_params; w->alpha = c->alpha; w->epsilon = c->epsilon; w->rmax_scale = c->rmax_scale;
    w->rmax = c->rmax; w->omega = c->omega; w->opt = c->opt; w->seed = c->seed;
    w->d_rw_idx = c->d_rw_idx; w->d_idx_off = c->d_idx_off; w->d_idx_cnt = c->d_idx_cnt;
    w->idx_len = c->idx_len; w->have_index = c->have_index;
    w->bk_scale = c->bk_scale; w->bk_scale_topk = c->bk_scale_topk;
    w->d_stamps = c->d_stamps;
    w->opt_ = c->opt_;
    w->balanced = c->balanced; w->bal_start = c->bal_start; w->c_pop = c->c_pop; w->c_edge = c->c_edge; w->t_walk = c->t_walk; w->t_idx = c->t_idx;
    w->batch_req = c->B; // same slot count as the first lane
    return FORA_OK;
}

int query_common(fora_ctx *c, const int32_t *sources, int nq, int with_idx, int flags, double *ppr_d,
                 uint64_t *ppr_fix, uint64_t *residue_fix, fora_query_stats *stats) {
    if (!c) return FORA_E_ARG;
    if (!c->n) return fail(c, FORA_E_ARG, "set_graph first");
    if (!c->have_params) return fail(c, FORA_E_ARG, "set_params first");
    if (nq < 0 || (nq && !sources)) return fail(c, FORA_E_ARG, "bad sources");
    if (with_idx && !c->have_index) return fail(c, FORA_E_ARG, "with_idx without an index (build or set one)");
    HIPCHK(c, hipSetDevice(c->device));
    for (int i = 0; i < nq; i++)
        if (sources[i] < 0 || sources[i] >= c->n) return fail(c, FORA_E_ARG, "source id out of range");
    const uint64_t n = (uint64_t)c->n;
    // A dangling source is its own whole answer (algo.h:961-965: reserve[s] = 1, rsum = 0, no push, no walks): it is
    // written here and never takes a slot.  (On the R-MAT variant with 52 % dangling nodes half of a batch's slots were
    // such sources, and every level launch and the tail kernel carried their empty workgroups: push of 1000 queries
    // 66.5 ms against 38 for the 483 others alone.)
    std::vector<int32_t> live_src;
    std::vector<int> live_at; // position of live source i in the caller's arrays
    for (int i = 0; i < nq; i++) {
        const int32_t s = sources[i];
        if (c->h_row_ptr[s + 1] != c->h_row_ptr[s]) { live_src.push_back(s); live_at.push_back(i); continue; }
        if (stats) {
            fora_query_stats &o = stats[i];
            memset(&o, 0, sizeof(o));
            o.ppr_sum_fix = FIX_ONE; o.dangling_source = 1; o.rmax_used = c->rmax; o.push_rounds = 1;
        }
        if (ppr_d) { double *row = ppr_d + (uint64_t)i * n; memset(row, 0, n * 8); row[s] = 1.0; }
        if (ppr_fix) { uint64_t *row = ppr_fix + (uint64_t)i * n; memset(row, 0, n * 8); row[s] = FIX_ONE; }
        if (residue_fix) memset(residue_fix + (uint64_t)i * n, 0, n * 8);
    }
    const int nl = (int)live_src.size();
    if (nl == 0) return FORA_OK;
    c->bk_div = 1; // (queries with smaller buckets, measured: LJ-sized 350 -> 220 q/s -- the indexed walks' results overflow into direct atomics; Twitter-2010-sized: no change)
    int rc = ensure_workspace(c, nl, c->omega);
    if (rc) return rc;
    // second lane when there is more than one batch to run
    fora_ctx *lanes[2] = {c, c};
    if (nl > c->B && c->opt_.pipeline == 1) { // opt-in: measured no gain on ws (kernels time-slice)
        rc = sync_twin(c);
        if (rc) return rc;
        rc = ensure_workspace(c->twin, c->B, c->omega);
        if (rc) { c->err = c->twin->err; return rc; }
        if (c->twin->B >= c->B) lanes[1] = c->twin;
    }
    struct Pending { fora_ctx *lane; int b0, nb; };
    std::vector<Pending> inflight;
    std::vector<fora_query_stats> st_tmp;
    auto finish = [&](const Pending &p) -> int {
        int r = batch_finish(p.lane);
        if (r) { if (p.lane != c) c->err = p.lane->err; return r; }
        if (stats) {
            st_tmp.resize((size_t)p.nb);
            fill_stats(p.lane, p.nb, st_tmp.data());
            for (int i = 0; i < p.nb; i++) stats[live_at[(size_t)p.b0 + i]] = st_tmp[(size_t)i];
        }
        // slots i .. j - 1 of the batch whose places in the caller's arrays are consecutive too: one copy
        for (int i = 0; i < p.nb && (ppr_d || ppr_fix || residue_fix);) {
            int j = i + 1;
            while (j < p.nb && live_at[(size_t)p.b0 + j] == live_at[(size_t)p.b0 + j - 1] + 1) j++;
            const uint64_t at = (uint64_t)live_at[(size_t)p.b0 + i] * n, from = (uint64_t)i * n, cnt = (uint64_t)(j - i) * n;
            if (ppr_d) {
                // u64 and f64 have the same size: copy raw, convert in place on the host
                double *dst = ppr_d + at;
                HIPCHK(c, hipMemcpy(dst, p.lane->d_ppr + from, cnt * 8, hipMemcpyDeviceToHost));
                uint64_t *raw = (uint64_t *)dst;
                for (uint64_t x = 0; x < cnt; x++) {
                    uint64_t u = raw[x];
                    dst[x] = std::ldexp((double)u, -62);
                }
            }
            if (ppr_fix) HIPCHK(c, hipMemcpy(ppr_fix + at, p.lane->d_ppr + from, cnt * 8, hipMemcpyDeviceToHost));
            if (residue_fix) HIPCHK(c, hipMemcpy(residue_fix + at, p.lane->d_residue + from, cnt * 8, hipMemcpyDeviceToHost));
            i = j;
        }
        return FORA_OK;
    };
    int k = 0;
    const int per = even_batch(nl, c->B);
    for (int b0 = 0; b0 < nl; b0 += per, k++) {
        const int nb = std::min(per, nl - b0);
        fora_ctx *lane = lanes[k & 1];
        // the lane's previous batch must be drained before its workspace is reused
        for (size_t i = 0; i < inflight.size();) {
            if (inflight[i].lane == lane) {
                rc = finish(inflight[i]);
                if (rc) return rc;
                inflight.erase(inflight.begin() + (long)i);
            } else i++;
        }
        rc = batch_begin(lane, live_src.data() + b0, nb, with_idx != 0, flags);
        if (rc) { if (lane != c) c->err = lane->err; return rc; }
        inflight.push_back({lane, b0, nb});
    }
    for (const Pending &p : inflight) {
        rc = finish(p);
        if (rc) return rc;
    }
    return FORA_OK;
}

} // namespace

// Two device conditions are not caller errors and are answered by running the call again from scratch (results never
// depend on either): message buckets (and their overflow list) too small -- double the bucket capacity and re-plan the
// workspace; a team of k_push_team that waited too long for a member (its workgroups were not co-resident: another
// context's kernels held CUs) -- the next calls push with the bucketed kernels.
constexpr int TEAM_SUSPEND_CALLS = 8;
template <class F> int with_bucket_retry(fora_ctx *c, F call) {
    const uint32_t scale0 = c ? c->bk_scale : 1, scale0t = c ? c->bk_scale_topk : 1;
    bool team_retried = false;
    auto forget_attempt = [&](const fora_timing &t0) { // the failed attempt must leave no trace in the timings: drop its event pairs and counters
        (void)hipSetDevice(c->device);
        (void)hipStreamSynchronize(c->stream);
        c->ev_used = 0;
        c->timing = t0;
        c->pending_nq = 0;
        if (c->twin) {
            (void)hipStreamSynchronize(c->twin->stream);
            c->twin->ev_used = 0;
            c->twin->pending_nq = 0;
        }
    };
    auto set_scales = [&](uint32_t s, uint32_t st) {
        c->bk_scale = s; c->bk_scale_topk = st;
        if (c->twin) { c->twin->bk_scale = s; c->twin->bk_scale_topk = st; }
    };
    auto drop_enlarged_plan = [&]() { // the enlarged plan did not help: do not keep it
        if (c->bk_scale == scale0 && c->bk_scale_topk == scale0t) return;
        set_scales(scale0, scale0t);
        (void)hipSetDevice(c->device);
        free_workspace(c);
        if (c->twin) free_workspace(c->twin);
    };
    for (;;) {
        const fora_timing t0 = c ? c->timing : fora_timing{};
        const int rc = call();
        if (c && rc == FORA_OK && c->team_suspend > 0 && !team_retried) c->team_suspend--; // (a retried call has just started the count)
        if (!c || rc != FORA_E_OVERFLOW) {
            if (c && rc != FORA_OK) drop_enlarged_plan();
            return rc;
        }
        if (c->team_timeout_seen && !team_retried) {
            c->team_timeout_seen = false;
            c->team_suspend = TEAM_SUSPEND_CALLS;
            c->team_fallbacks++;
            team_retried = true;
            forget_attempt(t0);
            continue;
        }
        const bool bucket = c->bucket_overflow || (c->twin && c->twin->bucket_overflow);
        // the multiplier of the regime the call planned with (fora_ctx::bk_div is set by the call itself)
        const bool topk_regime = c->bk_div > 1;
        const uint32_t cur = topk_regime ? c->bk_scale_topk : c->bk_scale;
        if (!bucket || cur >= (1u << 16)) {
            drop_enlarged_plan();
            return rc;
        }
        set_scales(topk_regime ? c->bk_scale : c->bk_scale * 2, topk_regime ? c->bk_scale_topk * 2 : c->bk_scale_topk);
        c->bucket_retries++;
        c->bucket_overflow = false;
        forget_attempt(t0);
        if (c->twin) c->twin->bucket_overflow = false;
        free_workspace(c);
        if (c->twin) free_workspace(c->twin);
    }
}

// =============================================================================== C ABI
extern "C" {

int fora_hip_device_count(void) {
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess) return 0;
    return ndev;
}

int fora_hip_create(int device, fora_ctx **out) {
    if (!out) return FORA_E_ARG;
    *out = nullptr;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) return FORA_E_NOGPU;
    if (device < 0 || device >= ndev) return FORA_E_ARG;
    fora_ctx *c = new (std::nothrow) fora_ctx();
    if (!c) return FORA_E_NOMEM;
    c->device = device;
    if (hipSetDevice(device) != hipSuccess || hipGetDeviceProperties(&c->prop, device) != hipSuccess ||
        hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess) {
        delete c;
        return FORA_E_HIP;
    }
    if (strncmp(c->prop.gcnArchName, "gfx950", 6) != 0) {
        // kernels are compiled for gfx950 only
        (void)hipStreamDestroy(c->stream);
        delete c;
        return FORA_E_NOGPU;
    }
    if (hipMalloc(&c->d_stamps, 32 * sizeof(unsigned long long)) != hipSuccess ||
        hipMemset(c->d_stamps, 0, 32 * sizeof(unsigned long long)) != hipSuccess) {
        (void)hipStreamDestroy(c->stream);
        delete c;
        return FORA_E_NOMEM;
    }
    c->opt_ = tunables_from_env();
    c->profiling = c->opt_.profile != 0;
    if (c->opt_.grid > 0) c->grid_blocks = (int)c->opt_.grid;
    *out = c;
    return FORA_OK;
}

void fora_hip_destroy(fora_ctx *c) {
    if (!c) return;
    (void)hipSetDevice(c->device);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    if (c->twin) {
        fora_ctx *w = c->twin;
        if (w->stream) (void)hipStreamSynchronize(w->stream);
        free_workspace(w);
        for (auto &p : w->ev_pool) { (void)hipEventDestroy(p.a); (void)hipEventDestroy(p.b); }
        if (w->stream) (void)hipStreamDestroy(w->stream);
        delete w; // graph / index pointers are owned by c
        c->twin = nullptr;
    }
    free_workspace(c);
    free_index(c);
    free_graph(c);
    dfree(c->d_stamps);
    for (auto &p : c->ev_pool) { (void)hipEventDestroy(p.a); (void)hipEventDestroy(p.b); }
    if (c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
}

const char *fora_hip_last_error(fora_ctx *c) { return c ? c->err.c_str() : "null ctx"; }

int fora_hip_device_info(fora_ctx *c, char *arch, int arch_len, int *cus, uint64_t *hbm_bytes) {
    if (!c) return FORA_E_ARG;
    if (arch && arch_len > 0) { strncpy(arch, c->prop.gcnArchName, (size_t)arch_len - 1); arch[arch_len - 1] = 0; }
    if (cus) *cus = c->prop.multiProcessorCount;
    if (hbm_bytes) *hbm_bytes = (uint64_t)c->prop.totalGlobalMem;
    return FORA_OK;
}


// Hub pre-aggregation of the narrow push (Dev::col_hub): the `hubs` nodes of largest in-degree (ties: lower id), numbered in
// id order so that the hubs of a bin are a contiguous range, and a copy of col that names them by that number.
static int build_hub_copy(fora_ctx *c, const int64_t *row_ptr, const int32_t *col) {
    (void)row_ptr;
    dfree(c->d_col_hub); dfree(c->d_hub_node); dfree(c->d_hub_first); c->hubs = 0; // (a rebuild: ensure_workspace)
    const int32_t n = c->n;
    const int64_t nnz = c->nnz;
    const int64_t wide_auto = nnz <= (1ll << 28) ? 2048 : 0;
    // the hub sums live in the bin kernel's dynamic LDS next to its static arrays: 48 KB in the narrow and the 512-thread
    // wide kernel, 32 KB in the 1024-thread one (its stage of 12 edges per thread takes 122 of the 160 KB)
    const uint64_t nbins_all = bins_of(c);
    const int64_t lds_cap = !want_wide(c) ? 6144 : nbins_all > (uint64_t)MAX_BINS_WIDE ? 4096 : 6144;
    // 4096 only when this graph really takes the team path (ensure_team has built its tables: then the bin kernel never runs
    // and k_push_tail is the copy's only reader); a graph the team path rejects pushes with the bin kernel, which wants 1024
    const bool for_team = want_team(c) && c->team_T != 0;
    c->hub_for_team = for_team;
    const int64_t narrow_auto = for_team ? 4096 : 1024;
    const int64_t want = std::min<int64_t>(std::max<int64_t>(want_wide(c) ? (c->opt_.hubs_wide < 0 ? wide_auto : c->opt_.hubs_wide) : (c->opt_.hubs < 0 ? narrow_auto : c->opt_.hubs), 0), lds_cap);
    if (want == 0 || nnz == 0 || c->opt_.direct == 1) return FORA_OK;
    if (want_wide(c) && (int64_t)nbins_all > (int64_t)want_pass_bins(c, (int)nbins_all)) return FORA_OK; // several bin passes per level: the passes read the row-sorted copy, hubs are never used (make_dev)
    std::vector<uint32_t> indeg((size_t)n, 0);
    for (int64_t e = 0; e < nnz; e++) indeg[(size_t)col[e]]++;
    std::vector<uint32_t> order((size_t)n);
    for (int32_t v = 0; v < n; v++) order[(size_t)v] = (uint32_t)v;
    const size_t H = (size_t)std::min<int64_t>(want, n);
    std::partial_sort(order.begin(), order.begin() + (long)H, order.end(),
                      [&](uint32_t a, uint32_t b) { return indeg[a] != indeg[b] ? indeg[a] > indeg[b] : a < b; });
    std::vector<uint32_t> hub_node(order.begin(), order.begin() + (long)H);
    std::sort(hub_node.begin(), hub_node.end());
    std::vector<uint32_t> hub_of((size_t)n, 0xFFFFFFFFu);
    for (size_t h = 0; h < H; h++) hub_of[hub_node[h]] = (uint32_t)h;
    const int nbins = (int)bins_of(c);
    std::vector<uint32_t> first((size_t)nbins + 1, 0);
    for (size_t h = 0; h < H; h++) first[(hub_node[h] >> bin_shift(c)) + 1]++;
    for (int b = 0; b < nbins; b++) first[(size_t)b + 1] += first[(size_t)b];
    std::vector<int32_t> ch((size_t)nnz);
    for (int64_t e = 0; e < nnz; e++) {
        const uint32_t h = hub_of[(size_t)col[e]];
        ch[(size_t)e] = h == 0xFFFFFFFFu ? col[e] : (int32_t)(0x80000000u | h);
    }
    HIPCHK(c, hipMalloc(&c->d_col_hub, (size_t)nnz * 4));
    HIPCHK(c, hipMalloc(&c->d_hub_node, H * 4));
    HIPCHK(c, hipMalloc(&c->d_hub_first, first.size() * 4));
    HIPCHK(c, hipMemcpy(c->d_col_hub, ch.data(), (size_t)nnz * 4, hipMemcpyHostToDevice));
    HIPCHK(c, hipMemcpy(c->d_hub_node, hub_node.data(), H * 4, hipMemcpyHostToDevice));
    HIPCHK(c, hipMemcpy(c->d_hub_first, first.data(), first.size() * 4, hipMemcpyHostToDevice));
    c->hubs = (uint32_t)H;
    c->hub_shift = bin_shift(c);
    return FORA_OK;
}

// Quad-padded copies of col (and of the hub copy) for the wide bin kernel (Dev::col4), built on the device from what
// set_graph has uploaded; only for graphs that run the wide layout in one bin pass per level.
static int build_quad_copies(fora_ctx *c) {
    dfree(c->d_col4); dfree(c->d_col_hub4); dfree(c->d_rowinfo4); c->quads = 0;
    if (!want_binned(c) || !want_wide(c) || c->nnz == 0 || c->opt_.quads == 0) return FORA_OK;
    const uint64_t nbins_all = bins_of(c);
    if ((int64_t)nbins_all > (int64_t)want_pass_bins(c, (int)nbins_all)) return FORA_OK; // several passes per level: pass-split rows, edge by edge
    const size_t n = (size_t)c->n;
    std::vector<uint64_t> ri4(n);
    uint64_t q = 0;
    // (Round 6, measured and dropped: rows placed so that each touches as few 64-byte lines as its length allows -- a row that would
    // straddle one line more than ceil(quads / 4) started at the next line.  LJ-sized bin kernel 315.4 / 314.8 ms against 320.7 / 314.5
    // back to back, Twitter-2010-sized 593.6 / 592.9 against 594.5 / 585.6: what a quad load costs is not the lines its row touches.)
    for (size_t v = 0; v < n; v++) {
        const uint64_t dg = (uint64_t)(c->h_row_ptr[v + 1] - c->h_row_ptr[v]);
        ri4[v] = (q << 24) | std::min<uint64_t>(dg, DEG_SAT);
        q += (dg + 3) / 4;
    }
    if (q >= (1ull << 40)) return FORA_OK;
    // The copies are an optimisation (make_dev falls back to single-edge reads without them): they must never make
    // set_graph fail.  Not built when they would take more than a quarter of the free memory (the slots need it more);
    // an allocation that fails all the same leaves "no quads", not an error.
    const uint64_t qbytes = std::max<uint64_t>(1, q) * 16, need = n * 8 + qbytes * (c->d_col_hub ? 2 : 1);
    size_t fr = 0, tot = 0;
    HIPCHK(c, hipMemGetInfo(&fr, &tot));
    if (need > fr / 4) return FORA_OK;
    auto give_up = [&]() { (void)hipGetLastError(); dfree(c->d_col4); dfree(c->d_col_hub4); dfree(c->d_rowinfo4); c->quads = 0; return FORA_OK; };
    if (hipMalloc(&c->d_rowinfo4, n * 8) != hipSuccess) return give_up();
    HIPCHK(c, hipMemcpy(c->d_rowinfo4, ri4.data(), n * 8, hipMemcpyHostToDevice));
    if (hipMalloc(&c->d_col4, qbytes) != hipSuccess) return give_up();
    const unsigned grid = (unsigned)std::min<size_t>((n + BLOCK - 1) / BLOCK, 1u << 20);
    hipLaunchKernelGGL(k_pad_quads, dim3(grid), dim3(BLOCK), 0, c->stream, c->n, (const int64_t *)c->d_row_ptr, (const int32_t *)c->d_col,
                       (const uint64_t *)c->d_rowinfo4, c->d_col4);
    if (c->d_col_hub) {
        if (hipMalloc(&c->d_col_hub4, qbytes) != hipSuccess) return give_up();
        hipLaunchKernelGGL(k_pad_quads, dim3(grid), dim3(BLOCK), 0, c->stream, c->n, (const int64_t *)c->d_row_ptr, (const int32_t *)c->d_col_hub,
                           (const uint64_t *)c->d_rowinfo4, c->d_col_hub4);
    }
    HIPCHK(c, hipStreamSynchronize(c->stream));
    c->quads = q;
    return FORA_OK;
}

// Degree-grouped walk copy (WalkDG, fora_kernels.h) of graphs that run the narrow layout: H hub records + at most 255
// out-degree classes whose tables fit a workgroup's LDS share.  Graphs that do not qualify keep k_walk_online.
static int build_walk_dg(fora_ctx *c, const int64_t *row_ptr, const int32_t *col) {
    const int32_t n = c->n;
    const int64_t nnz = c->nnz;
    if (c->opt_.walk_dg == 0 || c->opt_.no_compact == 1 || nnz >= (1ll << 31) || nnz == 0) return FORA_OK;
    if (!((uint64_t)n <= (uint64_t)MAX_BINS * BIN_SIZE && (uint64_t)n <= (1ull << SEG_BITS))) return FORA_OK; // narrow layout only
    std::vector<uint32_t> order((size_t)n); // nodes by (out-degree descending, id ascending)
    for (int32_t v = 0; v < n; v++) order[(size_t)v] = (uint32_t)v;
    auto degree = [&](uint32_t v) { return (uint32_t)(row_ptr[v + 1] - row_ptr[v]); };
    std::stable_sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) { return degree(a) > degree(b); });
    // smallest hub count that leaves at most 255 distinct degrees behind it
    uint32_t H = 0, K = 0;
    for (uint32_t h : {256u, 512u, 1024u, 2048u, 4096u}) {
        if (c->opt_.dg_hubs > 0 && (int64_t)h < c->opt_.dg_hubs) continue;
        const uint32_t hh = std::min<uint32_t>(h, (uint32_t)n);
        uint32_t k = 0;
        for (size_t i = hh; i < (size_t)n; i++) if (i == hh || degree(order[i]) != degree(order[i - 1])) k++;
        if (k <= 255) { H = hh; K = k; break; }
    }
    if (H == 0 && !(n <= 256)) return FORA_OK;
    if (n <= 256) { H = (uint32_t)n; K = 0; }
    uint32_t ts = 6;
    while ((((uint64_t)n + 256ull * (1ull << ts)) >> ts) > 8192) ts++; // at most 8192 blocks (8 KB of LDS)
    const uint32_t blk = 1u << ts;
    const uint32_t nrec = H + K;
    std::vector<uint32_t> rec((size_t)3 * nrec, 0), perm((size_t)n), inv;
    uint32_t *first = rec.data(), *rdeg = rec.data() + nrec, *base = rec.data() + 2 * (size_t)nrec;
    std::vector<uint8_t> T;
    uint64_t edge = 0;
    uint32_t id = 0, zero_first = 0xFFFFFFFFu;
    for (uint32_t i = 0; i < H; i++) { // hubs: one record each
        const uint32_t v = order[i];
        perm[v] = id; first[i] = id; rdeg[i] = degree(v); base[i] = (uint32_t)edge;
        if (rdeg[i] == 0 && zero_first == 0xFFFFFFFFu) zero_first = id;
        edge += rdeg[i]; id++;
    }
    uint32_t k = 0;
    for (size_t i = H; i < (size_t)n;) { // classes, each padded to whole blocks
        size_t j = i;
        const uint32_t dg = degree(order[i]);
        while (j < (size_t)n && degree(order[j]) == dg) j++;
        const uint32_t r = H + k;
        first[r] = id; rdeg[r] = dg; base[r] = (uint32_t)edge;
        if (dg == 0 && zero_first == 0xFFFFFFFFu) zero_first = id;
        for (size_t t = i; t < j; t++) perm[order[t]] = id + (uint32_t)(t - i);
        const uint32_t cnt = (uint32_t)(j - i), padded = (cnt + blk - 1) / blk * blk;
        for (uint32_t b = 0; b < padded / blk; b++) T.push_back((uint8_t)k);
        edge += (uint64_t)cnt * dg;
        id += padded;
        i = j; k++;
    }
    const uint32_t np = id; // ids in use (with padding)
    if (zero_first == 0xFFFFFFFFu) zero_first = np;
    // every id from zero_first on must be dangling: degrees descend, so the zero class (if any) is the last one
    uint32_t bits = 1;
    while ((1ull << bits) < (uint64_t)np) bits++;
    const size_t lds = (size_t)(H + 1) * 8 + (size_t)4 * nrec * 4 + T.size() + 4;
    if (lds > 28 * 1024 || bits > 31) return FORA_OK; // static + dynamic LDS of k_walk_dg stay under 64 KB
    inv.assign((size_t)np, 0);
    for (int32_t v = 0; v < n; v++) inv[perm[(size_t)v]] = (uint32_t)v;
    const size_t words = (size_t)(((uint64_t)nnz * bits + 31) / 32) + 2;
    std::vector<uint32_t> pk(words, 0);
    uint64_t e = 0;
    for (uint32_t x = 0; x < np; x++) { // rows in copy-id order, file order inside a row
        const uint32_t v = inv[x];
        if (perm[v] != x) continue; // padding id
        for (int64_t f = row_ptr[v]; f < row_ptr[v + 1]; f++, e++) {
            const uint64_t at = e * bits;
            const uint64_t y = (uint64_t)perm[(size_t)col[f]] << (at & 31);
            pk[at >> 5] |= (uint32_t)y;
            pk[(at >> 5) + 1] |= (uint32_t)(y >> 32);
        }
    }
    while (T.size() & 3) T.push_back(0);
    if (T.empty()) T.assign(4, 0);
    HIPCHK(c, hipMalloc(&c->d_dg_perm, (size_t)n * 4));
    HIPCHK(c, hipMalloc(&c->d_dg_inv, (size_t)np * 4));
    HIPCHK(c, hipMalloc(&c->d_dg_colp, words * 4));
    HIPCHK(c, hipMalloc(&c->d_dg_rec, std::max<size_t>(1, rec.size()) * 4));
    HIPCHK(c, hipMalloc(&c->d_dg_T, T.size()));
    HIPCHK(c, hipMemcpy(c->d_dg_perm, perm.data(), (size_t)n * 4, hipMemcpyHostToDevice));
    HIPCHK(c, hipMemcpy(c->d_dg_inv, inv.data(), (size_t)np * 4, hipMemcpyHostToDevice));
    HIPCHK(c, hipMemcpy(c->d_dg_colp, pk.data(), words * 4, hipMemcpyHostToDevice));
    if (!rec.empty()) HIPCHK(c, hipMemcpy(c->d_dg_rec, rec.data(), rec.size() * 4, hipMemcpyHostToDevice));
    HIPCHK(c, hipMemcpy(c->d_dg_T, T.data(), T.size(), hipMemcpyHostToDevice));
    // bucket order of the ids behind the hubs: 64-id blocks dealt round-robin to nbx bins
    const uint32_t nblk64 = (np - H + 63) / 64;
    const uint32_t nbx = std::max<uint32_t>(2, (nblk64 + 127) / 128); // (2 at least: floor(2^32 / nbx) + 1 must fit 32 bits)
    std::vector<uint32_t> invb((size_t)nbx * BIN_SIZE, 0);
    for (uint32_t x = H; x < np; x++) {
        const uint32_t u = x - H, b64 = u >> 6;
        invb[((size_t)(b64 % nbx) << BIN_SHIFT) | ((b64 / nbx) << 6) | (u & 63u)] = inv[x];
    }
    if (nbx <= (uint32_t)MAX_BINS) {
        HIPCHK(c, hipMalloc(&c->d_dg_invb, invb.size() * 4));
        HIPCHK(c, hipMemcpy(c->d_dg_invb, invb.data(), invb.size() * 4, hipMemcpyHostToDevice));
    }
    WalkDG g{};
    g.invb = c->d_dg_invb; g.nbx = c->d_dg_invb ? nbx : 0; g.nbx_magic = (uint32_t)((1ull << 32) / nbx) + 1;
    g.perm = c->d_dg_perm; g.inv = c->d_dg_inv; g.colp = c->d_dg_colp; g.rec = c->d_dg_rec; g.T = c->d_dg_T;
    g.H = H; g.nrec = nrec; g.nblk = (uint32_t)T.size(); g.ts = ts; g.bits = bits; g.zero_first = zero_first;
    g.bits32 = (uint64_t)nnz * bits < (1ull << 32) ? 1 : 0;
    c->dg = g;
    return FORA_OK;
}

int fora_hip_set_graph(fora_ctx *c, int32_t n, int64_t m_attr, const int64_t *row_ptr, const int32_t *col) {
    if (!c) return FORA_E_ARG;
    if (n <= 0 || !row_ptr || row_ptr[0] != 0) return fail(c, FORA_E_ARG, "bad graph");
    const int64_t nnz = row_ptr[n];
    if (nnz < 0 || (nnz && !col) || nnz >= (1ll << 32)) return fail(c, FORA_E_ARG, "bad graph (nnz): at most 2^32 - 1 edges");
    for (int32_t v = 0; v < n; v++)
        if (row_ptr[v + 1] < row_ptr[v]) return fail(c, FORA_E_ARG, "row_ptr not monotone");
    for (int64_t e = 0; e < nnz; e++)
        if (col[e] < 0 || col[e] >= n) return fail(c, FORA_E_ARG, "edge target out of range"); // graph.h:155-156
    HIPCHK(c, hipSetDevice(c->device));
    free_workspace(c);
    if (c->twin) free_workspace(c->twin);
    free_index(c);
    free_graph(c);
    c->bk_scale = 1; c->bk_scale_topk = 1;
    std::vector<uint64_t> rowinfo((size_t)n);
    std::vector<uint32_t> deg((size_t)n);
    int64_t n_dangling = 0;
    for (int32_t v = 0; v < n; v++) {
        const uint64_t dg = (uint64_t)(row_ptr[v + 1] - row_ptr[v]);
        n_dangling += dg == 0;
        if (dg > 0xFFFFFFFFull) return fail(c, FORA_E_ARG, "out-degree over 2^32");
        deg[v] = (uint32_t)dg;
        rowinfo[v] = ((uint64_t)row_ptr[v] << 24) | std::min<uint64_t>(dg, DEG_SAT);
    }
    HIPCHK(c, hipMalloc(&c->d_row_ptr, ((size_t)n + 1) * 8));
    HIPCHK(c, hipMalloc(&c->d_col, std::max<size_t>(1, (size_t)nnz) * 4));
    HIPCHK(c, hipMalloc(&c->d_rowinfo, (size_t)n * 8));
    HIPCHK(c, hipMalloc(&c->d_deg, (size_t)n * 4));
    HIPCHK(c, hipMemcpy(c->d_row_ptr, row_ptr, ((size_t)n + 1) * 8, hipMemcpyHostToDevice));
    if (nnz) HIPCHK(c, hipMemcpy(c->d_col, col, (size_t)nnz * 4, hipMemcpyHostToDevice));
    HIPCHK(c, hipMemcpy(c->d_rowinfo, rowinfo.data(), (size_t)n * 8, hipMemcpyHostToDevice));
    HIPCHK(c, hipMemcpy(c->d_deg, deg.data(), (size_t)n * 4, hipMemcpyHostToDevice));
    if (nnz < (1ll << 31) && c->opt_.no_compact != 1) { // compact walk-step copy
        uint32_t bits = 1;
        while ((1ull << bits) < (uint64_t)n) bits++;
        if (bits > 31) bits = 31;
        std::vector<uint32_t> rp32((size_t)n + 1);
        for (int32_t v = 0; v <= n; v++) rp32[v] = (uint32_t)row_ptr[v];
        const size_t words = (size_t)(((uint64_t)nnz * bits + 31) / 32) + 2;
        std::vector<uint32_t> pk(words, 0);
        for (int64_t e = 0; e < nnz; e++) {
            const uint64_t at = (uint64_t)e * bits;
            const uint64_t x = (uint64_t)(uint32_t)col[e] << (at & 31);
            pk[at >> 5] |= (uint32_t)x;
            pk[(at >> 5) + 1] |= (uint32_t)(x >> 32);
        }
        HIPCHK(c, hipMalloc(&c->d_rp32, rp32.size() * 4));
        HIPCHK(c, hipMalloc(&c->d_colp, words * 4));
        HIPCHK(c, hipMemcpy(c->d_rp32, rp32.data(), rp32.size() * 4, hipMemcpyHostToDevice));
        HIPCHK(c, hipMemcpy(c->d_colp, pk.data(), words * 4, hipMemcpyHostToDevice));
        c->colbits = bits;
    }
    c->h_row_ptr.assign(row_ptr, row_ptr + n + 1);
    c->n = n; c->m_attr = m_attr; c->nnz = nnz;
    c->dangling_frac = (double)n_dangling / (double)n;
    if (int rc = build_walk_dg(c, row_ptr, col)) return rc;
    c->team_checked = false;
    if (int rc = ensure_team(c)) return rc; // (before the hub copy: its size depends on whether the team path takes this graph)
    if (int rc = build_hub_copy(c, row_ptr, col)) return rc;
    if (int rc = build_quad_copies(c)) return rc;
    return FORA_OK;
}

int fora_hip_set_params(fora_ctx *c, double alpha, double epsilon, double rmax_scale, int opt, uint64_t seed) {
    if (!c) return FORA_E_ARG;
    if (!c->n) return fail(c, FORA_E_ARG, "set_graph first");
    if (!(alpha > 0 && alpha < 1) || !(epsilon > 0) || !(rmax_scale >= 0)) return fail(c, FORA_E_ARG, "bad params");
    // graph.h:177-178 then algo.h:455-463, in the reference's operand order
    const double delta = 1.0 / c->n, pfail = 1.0 / c->n;
    const long long m = c->m_attr;
    double rmax = epsilon * sqrt(delta / 3 / m / log(2 / pfail));
    if (opt) rmax *= rmax_scale / (1 - alpha);
    else rmax *= rmax_scale;
    const double omega = (2 + epsilon) * log(2 / pfail) / delta / epsilon / epsilon;
    c->alpha = alpha; c->epsilon = epsilon; c->rmax_scale = rmax_scale; c->opt = opt ? 1 : 0; c->seed = seed;
    c->rmax = rmax; c->omega = omega; c->have_params = true;
    return FORA_OK;
}

int fora_hip_set_params_raw(fora_ctx *c, double alpha, double rmax, double omega, int opt, uint64_t seed) {
    if (!c) return FORA_E_ARG;
    if (!(alpha > 0 && alpha < 1) || !(rmax > 0) || !(omega >= 0)) return fail(c, FORA_E_ARG, "bad params");
    c->alpha = alpha; c->rmax = rmax; c->omega = omega; c->opt = opt ? 1 : 0; c->seed = seed;
    c->have_params = true;
    return FORA_OK;
}

int fora_hip_get_params(fora_ctx *c, double *rmax, double *omega) {
    if (!c || !c->have_params) return FORA_E_ARG;
    if (rmax) *rmax = c->rmax;
    if (omega) *omega = c->omega;
    return FORA_OK;
}

int fora_hip_set_batch(fora_ctx *c, int batch) {
    if (!c || batch < 0) return FORA_E_ARG;
    if (batch != c->batch_req) { (void)hipSetDevice(c->device); free_workspace(c); if (c->twin) free_workspace(c->twin); }
    c->batch_req = batch;
    return FORA_OK;
}
int fora_hip_get_batch(fora_ctx *c) { return c ? c->B : FORA_E_ARG; }

int fora_hip_set_option(fora_ctx *c, const char *name, int64_t value) {
    if (!c || !name) return FORA_E_ARG;
    if (!strcmp(name, "reset")) { // back to the defaults / environment of fora_hip_create
        (void)hipSetDevice(c->device);
        free_workspace(c);
        if (c->twin) free_workspace(c->twin);
        c->opt_ = tunables_from_env();
        c->team_suspend = 0; // (a time-out's back-off too)
        c->profiling = c->opt_.profile != 0;
        c->grid_blocks = c->opt_.grid > 0 ? (int)c->opt_.grid : 2048;
        if (c->twin) { c->twin->opt_ = c->opt_; c->twin->grid_blocks = c->grid_blocks; }
        return FORA_OK;
    }
    if (!TEST_PATHS && schedule_option(name) && value != (strcmp(name, "rounds") ? (strcmp(name, "round_div") ? 0 : 4) : 1))
        return fail(c, FORA_E_ARG, std::string("option ") + name + ": the schedule experiments are not compiled into this library (build with -DFORA_TEST_PATHS=1: libfora_hip_test.so)");
    for (const auto &o : OPTIONS)
        if (!strcmp(name, o.name)) {
            if (c->opt_.*(o.field) == value) return FORA_OK;
            c->opt_.*(o.field) = value;
            if (o.layout) { (void)hipSetDevice(c->device); free_workspace(c); if (c->twin) free_workspace(c->twin); }
            if (!strcmp(name, "profile")) c->profiling = value != 0;
            if (!strcmp(name, "grid")) { c->grid_blocks = value > 0 ? (int)value : 2048; if (c->twin) c->twin->grid_blocks = c->grid_blocks; }
            return FORA_OK;
        }
    return fail(c, FORA_E_ARG, std::string("unknown option ") + name);
}

int fora_hip_get_option(fora_ctx *c, const char *name, int64_t *value) {
    if (!name || !value) return FORA_E_ARG;
    // properties of the BUILD: answered without a context (and so without a GPU)
    if (!strcmp(name, "test_paths")) { *value = TEST_PATHS ? 1 : 0; return FORA_OK; } // 1: libfora_hip_test.so (schedule experiments compiled in)
    if (!strcmp(name, "diag_build")) { *value = FORA_DIAG_BUILD; return FORA_OK; }    // 1: a probe / stamp / fake build (fora_diag.h) -- never the shipped library
    if (!c) return FORA_E_ARG;
    // read-only state of the engine beside the knobs
    if (!strcmp(name, "bucket_retries")) { *value = (int64_t)c->bucket_retries; return FORA_OK; } // re-runs with doubled message buckets
    if (!strcmp(name, "team_fallbacks")) { *value = (int64_t)c->team_fallbacks; return FORA_OK; } // calls re-run with the bucketed push after a team time-out
    if (!strcmp(name, "team_suspended")) { *value = c->team_suspend; return FORA_OK; }             // calls left that do not try the team push
    if (!strcmp(name, "team_members")) { *value = c->team_T; return FORA_OK; }                      // 0: this graph / workspace has no team push
    if (!strcmp(name, "team_cooperative")) { *value = c->team_coop_ok && !c->team_coop_failed ? 1 : 0; return FORA_OK; }
    for (const auto &o : OPTIONS)
        if (!strcmp(name, o.name)) { *value = c->opt_.*(o.field); return FORA_OK; }
    return fail(c, FORA_E_ARG, std::string("unknown option ") + name);
}

int fora_hip_set_balanced(fora_ctx *c, int on, double start_scale, double c_pop, double c_edge, double t_walk, double t_idx) {
    if (!c) return FORA_E_ARG;
    c->balanced = on != 0;
    c->bal_start = start_scale > 0 ? start_scale : 8;
    c->c_pop = c_pop > 0 ? c_pop : 2.0e-11;
    c->c_edge = c_edge > 0 ? c_edge : 2.4e-11;
    c->t_walk = t_walk > 0 ? t_walk : 6.5e-11;
    c->t_idx = t_idx > 0 ? t_idx : 2.2e-11;
    return FORA_OK;
}

// ---- index ---------------------------------------------------------------------
static uint64_t host_index_sizes(const fora_ctx *c, uint64_t *off, uint64_t *cnt) {
    // build.h:325-334
    uint64_t total = 0;
    for (int32_t v = 0; v < c->n; v++) {
        const size_t deg = (size_t)(c->h_row_ptr[v + 1] - c->h_row_ptr[v]);
        unsigned long num_rw;
        if (c->opt) num_rw = (unsigned long)ceil(deg * c->rmax * (1 - c->alpha) * c->omega);
        else num_rw = (unsigned long)ceil(deg * c->rmax * c->omega);
        if (off) off[v] = total;
        if (cnt) cnt[v] = num_rw;
        total += num_rw;
    }
    return total;
}

int fora_hip_index_sizes(fora_ctx *c, uint64_t *total, uint64_t *off, uint64_t *cnt) {
    if (!c || !c->n || !c->have_params) return fail(c, FORA_E_ARG, "set_graph and set_params first");
    const uint64_t t = host_index_sizes(c, off, cnt);
    if (total) *total = t;
    return FORA_OK;
}

int fora_hip_build_index(fora_ctx *c) {
    if (!c || !c->n || !c->have_params) return fail(c, FORA_E_ARG, "set_graph and set_params first");
    HIPCHK(c, hipSetDevice(c->device));
    std::vector<uint64_t> off((size_t)c->n), cnt((size_t)c->n);
    const uint64_t total = host_index_sizes(c, off.data(), cnt.data());
    free_index(c);
    HIPCHK(c, hipMalloc(&c->d_rw_idx, std::max<uint64_t>(1, total) * 4));
    HIPCHK(c, hipMalloc(&c->d_idx_off, (size_t)c->n * 8));
    HIPCHK(c, hipMalloc(&c->d_idx_cnt, (size_t)c->n * 8));
    HIPCHK(c, hipMemcpy(c->d_idx_off, off.data(), (size_t)c->n * 8, hipMemcpyHostToDevice));
    HIPCHK(c, hipMemcpy(c->d_idx_cnt, cnt.data(), (size_t)c->n * 8, hipMemcpyHostToDevice));
    c->idx_len = total;
    c->bk_div = 1;
    int rc = ensure_workspace(c, 1, (double)total);
    if (rc) return rc;
    Dev d = make_dev(c, 1, true);
    HIPCHK(c, hipMemsetAsync(c->d_counters, 0, N_COUNTERS * sizeof(unsigned long long), c->stream));
    HIPCHK(c, hipMemsetAsync(c->d_err, 0, sizeof(uint32_t), c->stream));
    HIPCHK(c, hipMemsetAsync(c->d_wit_count, 0, (size_t)c->B * 4 * CSTRIDE, c->stream));
    const uint32_t chunks = (uint32_t)std::min<int64_t>(((int64_t)c->n + BLOCK - 1) / BLOCK, 2048);
    int h = ev_begin(c, 4);
    hipLaunchKernelGGL(k_index_alloc, dim3(chunks), dim3(BLOCK), 0, c->stream, d);
    ev_end(c, h);
    h = ev_begin(c, 3);
    hipLaunchKernelGGL(k_walk_online<WALK_TO_INDEX>, dim3(walk_grid_x(c, 1), 1), dim3(BLOCK), 0, c->stream, d, 0u,
                       c->opt ? 1 : 0, c->d_rw_idx);
    ev_end(c, h);
    rc = check_dev_err(c);
    if (rc) return rc;
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(c, FORA_E_HIP, std::string("build_index: ") + hipGetErrorString(e));
    ev_collect(c);
    c->have_index = true;
    return FORA_OK;
}

int fora_hip_get_index(fora_ctx *c, int32_t *rw_idx, uint64_t len, uint64_t *off, uint64_t *cnt) {
    if (!c || !c->have_index) return fail(c, FORA_E_ARG, "no index");
    if (rw_idx && len < c->idx_len) return fail(c, FORA_E_ARG, "rw_idx buffer too small");
    HIPCHK(c, hipSetDevice(c->device));
    if (rw_idx && c->idx_len) HIPCHK(c, hipMemcpy(rw_idx, c->d_rw_idx, c->idx_len * 4, hipMemcpyDeviceToHost));
    if (off) HIPCHK(c, hipMemcpy(off, c->d_idx_off, (size_t)c->n * 8, hipMemcpyDeviceToHost));
    if (cnt) HIPCHK(c, hipMemcpy(cnt, c->d_idx_cnt, (size_t)c->n * 8, hipMemcpyDeviceToHost));
    return FORA_OK;
}

int fora_hip_set_index(fora_ctx *c, const int32_t *rw_idx, uint64_t len, const uint64_t *off, const uint64_t *cnt) {
    if (!c || !c->n) return fail(c, FORA_E_ARG, "set_graph first");
    if (!off || !cnt || (len && !rw_idx)) return fail(c, FORA_E_ARG, "bad index");
    for (int32_t v = 0; v < c->n; v++)
        if (off[v] + cnt[v] > len) return fail(c, FORA_E_ARG, "index entry range out of bounds");
    for (uint64_t i = 0; i < len; i++)
        if (rw_idx[i] < 0 || rw_idx[i] >= c->n) return fail(c, FORA_E_ARG, "index endpoint out of range");
    HIPCHK(c, hipSetDevice(c->device));
    free_index(c);
    HIPCHK(c, hipMalloc(&c->d_rw_idx, std::max<uint64_t>(1, len) * 4));
    HIPCHK(c, hipMalloc(&c->d_idx_off, (size_t)c->n * 8));
    HIPCHK(c, hipMalloc(&c->d_idx_cnt, (size_t)c->n * 8));
    if (len) HIPCHK(c, hipMemcpy(c->d_rw_idx, rw_idx, len * 4, hipMemcpyHostToDevice));
    HIPCHK(c, hipMemcpy(c->d_idx_off, off, (size_t)c->n * 8, hipMemcpyHostToDevice));
    HIPCHK(c, hipMemcpy(c->d_idx_cnt, cnt, (size_t)c->n * 8, hipMemcpyHostToDevice));
    c->idx_len = len;
    c->have_index = true;
    return FORA_OK;
}

int fora_hip_clear_index(fora_ctx *c) {
    if (!c) return FORA_E_ARG;
    (void)hipSetDevice(c->device);
    free_index(c);
    return FORA_OK;
}

// ---- queries -------------------------------------------------------------------
int fora_hip_query_batch(fora_ctx *c, const int32_t *sources, int nq, int with_idx, double *ppr_out,
                         fora_query_stats *stats) {
    return with_bucket_retry(c, [&] { return query_common(c, sources, nq, with_idx, 0, ppr_out, nullptr, nullptr, stats); });
}

int fora_hip_query_batch_fix(fora_ctx *c, const int32_t *sources, int nq, int with_idx, uint64_t *ppr_fix_out,
                             uint64_t *residue_fix_out, fora_query_stats *stats) {
    return with_bucket_retry(c, [&] { return query_common(c, sources, nq, with_idx, 0, nullptr, ppr_fix_out, residue_fix_out, stats); });
}

int fora_hip_push_batch(fora_ctx *c, const int32_t *sources, int nq, uint64_t *reserve_fix_out,
                        uint64_t *residue_fix_out, fora_query_stats *stats) {
    return with_bucket_retry(c, [&] { return query_common(c, sources, nq, 0, RUN_PUSH_ONLY, nullptr, reserve_fix_out, residue_fix_out, stats); });
}

int fora_hip_walk_counts(fora_ctx *c, const double *residue, double rsum, uint64_t *num_s_rw, uint64_t *n_rw) {
    if (!c || !c->n || !c->have_params || !residue || !num_s_rw) return fail(c, FORA_E_ARG, "bad call");
    HIPCHK(c, hipSetDevice(c->device));
    double *d_r = nullptr;
    uint64_t *d_num = nullptr, *d_n = nullptr;
    const size_t n = (size_t)c->n;
    HIPCHK(c, hipMalloc(&d_r, n * 8));
    HIPCHK(c, hipMalloc(&d_num, n * 8));
    HIPCHK(c, hipMalloc(&d_n, 8));
    HIPCHK(c, hipMemcpy(d_r, residue, n * 8, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k_walk_counts_f64, dim3((unsigned)((n + BLOCK - 1) / BLOCK)), dim3(BLOCK), 0, c->stream,
                       c->n, d_r, rsum, c->omega, c->alpha, c->opt, d_num, d_n);
    HIPCHK(c, hipStreamSynchronize(c->stream));
    HIPCHK(c, hipMemcpy(num_s_rw, d_num, n * 8, hipMemcpyDeviceToHost));
    uint64_t N = 0;
    HIPCHK(c, hipMemcpy(&N, d_n, 8, hipMemcpyDeviceToHost));
    if (n_rw) *n_rw = N;
    dfree(d_r); dfree(d_num); dfree(d_n);
    return FORA_OK;
}

int fora_hip_walks(fora_ctx *c, uint32_t stream_id, uint32_t round, int no_zero_hop, const int32_t *starts,
                   const uint64_t *js, int64_t count, int32_t *dests) {
    if (!c || !c->n || !c->have_params || count < 0) return fail(c, FORA_E_ARG, "bad call");
    if (count == 0) return FORA_OK;
    for (int64_t i = 0; i < count; i++)
        if (starts[i] < 0 || starts[i] >= c->n) return fail(c, FORA_E_ARG, "walk start out of range");
    HIPCHK(c, hipSetDevice(c->device));
    int32_t *d_s = nullptr, *d_d = nullptr;
    uint64_t *d_j = nullptr;
    HIPCHK(c, hipMalloc(&d_s, (size_t)count * 4));
    HIPCHK(c, hipMalloc(&d_d, (size_t)count * 4));
    HIPCHK(c, hipMalloc(&d_j, (size_t)count * 8));
    HIPCHK(c, hipMemcpy(d_s, starts, (size_t)count * 4, hipMemcpyHostToDevice));
    HIPCHK(c, hipMemcpy(d_j, js, (size_t)count * 8, hipMemcpyHostToDevice));
    Dev d{};
    d.n = c->n; d.rowinfo = c->d_rowinfo; d.row_ptr = c->d_row_ptr; d.col = c->d_col;
    d.alpha32 = (uint32_t)(c->alpha * 4294967296.0);
    d.seed_lo = (uint32_t)c->seed; d.seed_hi = (uint32_t)(c->seed >> 32);
    hipLaunchKernelGGL(k_walks_raw, dim3((unsigned)((count + BLOCK - 1) / BLOCK)), dim3(BLOCK), 0, c->stream, d,
                       stream_id, round, no_zero_hop, d_s, d_j, count, d_d);
    HIPCHK(c, hipStreamSynchronize(c->stream));
    HIPCHK(c, hipMemcpy(dests, d_d, (size_t)count * 4, hipMemcpyDeviceToHost));
    dfree(d_s); dfree(d_d); dfree(d_j);
    return FORA_OK;
}

// top-k driver: fora_query_topk_new (query.h:972-1045) for a batch of slots.  All active slots
// are in the same round, so delta / rmax / omega are uniform per round; finished slots drop out.
static int topk_batch_impl(fora_ctx *c, const int32_t *sources, int nq, int k, double epsilon, double rmax_scale,
                           int with_idx, int32_t *ids, double *scores, int32_t *rounds);
// The index cursors of a new batch all read as zero: a new epoch (the slabs themselves are cleared when they are allocated
// and when the 24-bit epoch wraps).
static int next_cursor_epoch(fora_ctx *c) {
    if (c->cursor_epoch == 0 || c->cursor_epoch >= 0xFFFFFFu) {
        HIPCHK(c, hipMemsetAsync(c->d_cursor, 0, (uint64_t)c->B * (uint64_t)c->n * 8, c->stream));
        c->cursor_epoch = 0;
    }
    c->cursor_epoch++;
    return FORA_OK;
}

int fora_hip_topk_batch(fora_ctx *c, const int32_t *sources, int nq, int k, double epsilon, double rmax_scale,
                        int with_idx, int32_t *ids, double *scores, int32_t *rounds) {
    return with_bucket_retry(c, [&] { return topk_batch_impl(c, sources, nq, k, epsilon, rmax_scale, with_idx, ids, scores, rounds); });
}
static int topk_batch_impl(fora_ctx *c, const int32_t *sources, int nq, int k, double epsilon, double rmax_scale,
                           int with_idx, int32_t *ids, double *scores, int32_t *rounds) {
    if (!c) return FORA_E_ARG;
    if (!c->n) return fail(c, FORA_E_ARG, "set_graph first");
    if (!c->have_params) return fail(c, FORA_E_ARG, "set_params first (alpha, seed)");
    if (k == 0) k = 500; // query.h:975
    if (nq < 0 || (nq && (!sources || !ids || !scores))) return fail(c, FORA_E_ARG, "bad arguments");
    if (k < 2 || k >= c->n - 1) return fail(c, FORA_E_ARG, "k out of range (query.h:1317-1318)");
    if (k > SEL_MAXK) return fail(c, FORA_E_ARG, "k > 1024 not supported");
    if (!(epsilon > 0) || !(rmax_scale >= 0)) return fail(c, FORA_E_ARG, "bad epsilon / rmax_scale");
    if (with_idx && !c->have_index) return fail(c, FORA_E_ARG, "with_idx without an index");
    for (int i = 0; i < nq; i++)
        if (sources[i] < 0 || sources[i] >= c->n) return fail(c, FORA_E_ARG, "source id out of range");
    HIPCHK(c, hipSetDevice(c->device));
    const double n_d = (double)c->n;
    const double min_delta = 1.0 / c->n;           // query.h:974
    const double init_delta = 1.0 / k / 10;        // query.h:976
    const double pfail = 1.0 / c->n / c->n;        // query.h:977
    const long long m = c->m_attr;
    (void)n_d;
    if (!(init_delta >= min_delta)) { // k > n/10: the reference's round loop (query.h:1001) never runs, topk_ppr sees an empty ppr
        for (size_t i = 0; i < (size_t)nq * k; i++) { ids[i] = 0; scores[i] = 0.0; }
        if (rounds) for (int i = 0; i < nq; i++) rounds[i] = 0;
        return FORA_OK;
    }
    // omega of the last possible round bounds the walk work list
    const double omega_max = (2 + epsilon) * log(2 / pfail) / min_delta / epsilon / epsilon;
    c->bk_div = want_wide(c) && c->opt_.bkcap <= 0 ? (uint32_t)std::max<int64_t>(1, c->opt_.topk_bk_div) : 1u;
    int rc = ensure_workspace(c, nq, omega_max);
    if (rc) return rc;
    const uint64_t n = (uint64_t)c->n;
    if (!c->d_ppr2) {
        const uint64_t slab = (uint64_t)c->B * n;
        HIPCHK(c, hipMalloc(&c->d_ppr2, slab * 8));
        HIPCHK(c, hipMalloc(&c->d_cursor, slab * 8));
        HIPCHK(c, hipMalloc(&c->d_active, (size_t)c->B));
        HIPCHK(c, hipMalloc(&c->d_above, (size_t)c->B * 8));
    }
    if (c->topk_cap < c->B * k) {
        dfree(c->d_topk_ids); dfree(c->d_topk_sc);
        HIPCHK(c, hipMalloc(&c->d_topk_ids, (size_t)c->B * k * 4));
        HIPCHK(c, hipMalloc(&c->d_topk_sc, (size_t)c->B * k * 8));
        c->topk_cap = c->B * k;
    }
    const uint32_t chunks = slab_grid_x(c, std::min(nq, c->B));
    std::vector<uint8_t> active, inactive;
    std::vector<unsigned long long> above;
    const int per = even_batch(nq, c->B);
    for (int b0 = 0; b0 < nq; b0 += per) {
        const int nb = std::min(per, nq - b0);
        const int hb = ev_begin(c, 5);
        rc = reset_batch_state(c, nb, sources + b0);
        if (rc) return rc;
        if (with_idx) { int rce = next_cursor_epoch(c); if (rce) return rce; } // query.h:997-998: every cursor of the batch reads as 0
        Dev d = make_dev(c, nb, with_idx != 0);
        hipLaunchKernelGGL(k_init_batch, dim3((nb + BLOCK - 1) / BLOCK), dim3(BLOCK), 0, c->stream, d, 1);
        active.assign((size_t)nb, 1);
        for (int i = 0; i < nb; i++) // dangling source: query.h:1007-1011, one round, ppr = e_s
            if (c->h_row_ptr[sources[b0 + i] + 1] == c->h_row_ptr[sources[b0 + i]]) active[i] = 0;
        std::vector<int32_t> nround((size_t)nb, 1);
        std::vector<double> sel_thr((size_t)nb, 0.0); // slots that stop with k entries >= T: the top k are among those
        // ppr2 := reserve once for the slots that never run a round (dangling sources: ppr = e_s); every other slot's ppr2 is written
        // by its first round's copy (round 5 copied all slots here: one 12-GB slab pass per Twitter-2010-sized batch for nothing)
        inactive.assign((size_t)nb, 0);
        bool any_inactive = false;
        for (int i = 0; i < nb; i++) { inactive[i] = active[i] ? 0 : 1; any_inactive |= inactive[i] != 0; }
        if (any_inactive) {
            HIPCHK(c, hipMemcpyAsync(c->d_active, inactive.data(), (size_t)nb, hipMemcpyHostToDevice, c->stream));
            hipLaunchKernelGGL(k_copy_slab, dim3(chunks, nb), dim3(BLOCK), 0, c->stream, c->n, c->d_ppr, c->d_ppr2, (const uint8_t *)c->d_active);
        }
        double delta = init_delta;
        int round = 0;
        while (delta >= min_delta) { // query.h:1001
            bool any = false;
            for (int i = 0; i < nb; i++) any |= active[i] != 0;
            if (!any) break;
            round++;
            // fora_topk_setting, algo.h:466-474
            double rmax = epsilon * sqrt(delta / 3 / m / log(2 / pfail));
            rmax *= sqrt(1.0 * m * rmax) * rmax_scale * 3;
            const double omega = (2 + epsilon) * log(2 / pfail) / delta / epsilon / epsilon;
            for (int i = 0; i < nb; i++) if (active[i]) nround[i] = round;
            HIPCHK(c, hipMemcpyAsync(c->d_active, active.data(), (size_t)nb, hipMemcpyHostToDevice, c->stream));
            HIPCHK(c, hipMemsetAsync(c->d_counters, 0, N_COUNTERS * sizeof(unsigned long long), c->stream));
            HIPCHK(c, hipMemsetAsync(c->d_above, 0, (size_t)nb * 8, c->stream));
            HIPCHK(c, hipMemsetAsync(c->d_wit_count, 0, (size_t)c->B * 4 * CSTRIDE, c->stream));
            rc = reset_binned_counters(c);
            if (rc) return rc;
            d = make_dev(c, nb, with_idx != 0, rmax, omega);
            int h = ev_begin(c, 4);
            hipLaunchKernelGGL(k_topk_frontier, dim3(chunks, nb), dim3(BLOCK), 0, c->stream, d, (const uint8_t *)c->d_active);
            ev_end(c, h);
            rc = run_push_levels(c, d, nullptr, 0, true); // algo.h:1020-1093
            if (rc) return rc;
            // compute_ppr_with_fwdidx_topk, query.h:521-636, into ppr2
            h = ev_begin(c, 4);
            hipLaunchKernelGGL(k_copy_slab, dim3(chunks, nb), dim3(BLOCK), 0, c->stream, c->n, c->d_ppr, c->d_ppr2,
                               (const uint8_t *)c->d_active);
            ev_end(c, h);
            Dev dw = d;
            dw.ppr = c->d_ppr2;
            h = ev_begin(c, 2);
            hipLaunchKernelGGL(k_walk_alloc<ALLOC_TOPK>, (dw.wide && (dw.slot_major & 8u)) ? dim3(nb, chunks) : dim3(chunks, nb), dim3(BLOCK), 0, c->stream, dw, with_idx ? 1 : 0,
                               (const uint8_t *)c->d_active, c->d_cursor, (unsigned long long *)nullptr, c->cursor_epoch);
            ev_end(c, h);
            launch_walks(c, dw, nb, with_idx != 0, (uint32_t)round, with_idx ? 1 : 0);
            const double T = (1 + epsilon) * delta; // query.h:1030
            h = ev_begin(c, 4);
            hipLaunchKernelGGL(k_count_above, dim3(std::min<uint32_t>(chunks, 256), nb), dim3(BLOCK), 0, c->stream, dw,
                               (const uint8_t *)c->d_active, T, c->d_above);
            ev_end(c, h);
            above.assign((size_t)nb, 0);
            HIPCHK(c, hipMemcpyAsync(above.data(), c->d_above, (size_t)nb * 8, hipMemcpyDeviceToHost, c->stream));
            rc = check_dev_err(c);
            if (rc) return rc;
            for (int i = 0; i < nb; i++)
                if (active[i] && (above[i] >= (unsigned long long)k || delta <= min_delta)) {
                    active[i] = 0;
                    if (above[i] >= (unsigned long long)k) sel_thr[i] = T;
                }
            if (delta <= min_delta) break;
            delta = std::max(min_delta, delta / 4.0); // query.h:1041
        }
        // topk_ppr, algo.h:592-610
        Dev ds = make_dev(c, nb, false);
        ds.ppr = c->d_ppr2;
        int h = ev_begin(c, 4);
        rc = launch_select(c, ds, nb, k, c->d_topk_ids, c->d_topk_sc, 0, sel_thr.data());
        if (rc) return rc;
        ev_end(c, h);
        ev_end(c, hb);
        HIPCHK(c, hipMemcpyAsync(ids + (size_t)b0 * k, c->d_topk_ids, (size_t)nb * k * 4, hipMemcpyDeviceToHost, c->stream));
        HIPCHK(c, hipMemcpyAsync(scores + (size_t)b0 * k, c->d_topk_sc, (size_t)nb * k * 8, hipMemcpyDeviceToHost, c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));
        hipError_t e = hipGetLastError();
        if (e != hipSuccess) return fail(c, FORA_E_HIP, std::string("topk: ") + hipGetErrorString(e));
        ev_collect(c);
        HIPCHK(c, hipMemcpy(c->h_qs.data(), c->d_qs, (size_t)nb * sizeof(QState), hipMemcpyDeviceToHost));
        for (int i = 0; i < nb; i++) { // counters accumulated over all rounds of the slot
            c->timing.pops += c->h_qs[i].pops;
            c->timing.relax += c->h_qs[i].relax;
            c->timing.walks += c->h_qs[i].n_walks;
            c->timing.idx_hits += c->h_qs[i].n_hit;
        }
        if (rounds) for (int i = 0; i < nb; i++) rounds[b0 + i] = nround[i];
    }
    return FORA_OK;
}

// top-k with bounds: fora_query_topk_with_bound (query.h:909-969) for a batch of slots.  As in the --opt driver all
// active slots share a round (delta halves per round), finished slots drop out.  zero_ppr_upper_bound (query.h:935,
// :748) only ever feeds itself in the reference and is not kept.
static int topk_bound_batch_impl(fora_ctx *c, const int32_t *sources, int nq, int k, double epsilon, double rmax_scale,
                                 double ppr_decay_alpha, int with_idx, int32_t *ids, double *scores, int32_t *rounds);
int fora_hip_topk_bound_batch(fora_ctx *c, const int32_t *sources, int nq, int k, double epsilon, double rmax_scale,
                              double ppr_decay_alpha, int with_idx, int32_t *ids, double *scores, int32_t *rounds) {
    return with_bucket_retry(c, [&] {
        return topk_bound_batch_impl(c, sources, nq, k, epsilon, rmax_scale, ppr_decay_alpha, with_idx, ids, scores, rounds);
    });
}
static int topk_bound_batch_impl(fora_ctx *c, const int32_t *sources, int nq, int k, double epsilon, double rmax_scale,
                                 double ppr_decay_alpha, int with_idx, int32_t *ids, double *scores, int32_t *rounds) {
    if (!c) return FORA_E_ARG;
    if (!c->n) return fail(c, FORA_E_ARG, "set_graph first");
    if (!c->have_params) return fail(c, FORA_E_ARG, "set_params first (alpha, seed)");
    if (nq < 0 || (nq && (!sources || !ids || !scores))) return fail(c, FORA_E_ARG, "bad arguments");
    if (k < 2 || k >= c->n - 1) return fail(c, FORA_E_ARG, "k out of range (query.h:1317-1318)");
    if (k > SEL_MAXK) return fail(c, FORA_E_ARG, "k > 1024 not supported");
    if (!(epsilon > 0) || !(rmax_scale >= 0)) return fail(c, FORA_E_ARG, "bad epsilon / rmax_scale");
    if (!(ppr_decay_alpha > 0 && ppr_decay_alpha < 1)) return fail(c, FORA_E_ARG, "bad ppr_decay_alpha");
    if (with_idx && !c->have_index) return fail(c, FORA_E_ARG, "with_idx without an index");
    for (int i = 0; i < nq; i++)
        if (sources[i] < 0 || sources[i] >= c->n) return fail(c, FORA_E_ARG, "source id out of range");
    HIPCHK(c, hipSetDevice(c->device));
    const double min_delta = 1.0 / c->n;                                                                    // query.h:911
    const double init_delta = 1.0 / 4;                                                                      // :912
    const double threshold = (1.0 - ppr_decay_alpha) / pow(500, ppr_decay_alpha) / pow(c->n, 1 - ppr_decay_alpha); // :913
    const double pfail = 1.0 / c->n / c->n / log(c->n);                                                     // :915
    const double L = log(2 / pfail);
    const long long m = c->m_attr;
    const double omega_max = (2 + epsilon) * L / min_delta / epsilon / epsilon;
    c->bk_div = 1;
    int rc = ensure_workspace(c, nq, omega_max);
    if (rc) return rc;
    const uint64_t n = (uint64_t)c->n;
    const uint64_t slab = (uint64_t)c->B * n;
    if (!c->d_ppr2) {
        HIPCHK(c, hipMalloc(&c->d_ppr2, slab * 8));
        HIPCHK(c, hipMalloc(&c->d_cursor, slab * 8));
        HIPCHK(c, hipMalloc(&c->d_active, (size_t)c->B));
        HIPCHK(c, hipMalloc(&c->d_above, (size_t)c->B * 8));
    }
    if (!c->d_upper) {
        HIPCHK(c, hipMalloc(&c->d_upper, slab * 8));
        HIPCHK(c, hipMalloc(&c->d_lower, slab * 8));
        HIPCHK(c, hipMalloc(&c->d_filter, slab));
        HIPCHK(c, hipMemset(c->d_filter, 0, slab));
        HIPCHK(c, hipMalloc(&c->d_fail, (size_t)c->B * 4));
        HIPCHK(c, hipMalloc(&c->d_round_walks, (size_t)c->B * 8));
    }
    if (c->topk_cap < c->B * k) {
        dfree(c->d_topk_ids); dfree(c->d_topk_sc);
        HIPCHK(c, hipMalloc(&c->d_topk_ids, (size_t)c->B * k * 4));
        HIPCHK(c, hipMalloc(&c->d_topk_sc, (size_t)c->B * k * 8));
        c->topk_cap = c->B * k;
    }
    if (c->lb_cap < c->B * k) {
        dfree(c->d_lb_ids); dfree(c->d_lb_sc);
        HIPCHK(c, hipMalloc(&c->d_lb_ids, (size_t)c->B * k * 4));
        HIPCHK(c, hipMalloc(&c->d_lb_sc, (size_t)c->B * k * 8));
        c->lb_cap = c->B * k;
    }
    const uint32_t chunks = slab_grid_x(c, std::min(nq, c->B));
    std::vector<uint8_t> active, inactive;
    std::vector<unsigned long long> above;
    std::vector<uint32_t> failv;
    const int per = even_batch(nq, c->B);
    for (int b0 = 0; b0 < nq; b0 += per) {
        const int nb = std::min(per, nq - b0);
        const int hb = ev_begin(c, 5);
        rc = reset_batch_state(c, nb, sources + b0);
        if (rc) return rc;
        if (with_idx) { int rce = next_cursor_epoch(c); if (rce) return rce; } // query.h:937-938: every cursor of the batch reads as 0
        hipLaunchKernelGGL(k_bounds_reset, dim3(chunks, nb), dim3(BLOCK), 0, c->stream, c->n, c->d_upper, c->d_lower); // :941-942
        Dev d = make_dev(c, nb, with_idx != 0);
        hipLaunchKernelGGL(k_init_batch, dim3((nb + BLOCK - 1) / BLOCK), dim3(BLOCK), 0, c->stream, d, 1);
        active.assign((size_t)nb, 1);
        for (int i = 0; i < nb; i++) // dangling source: query.h:951-955
            if (c->h_row_ptr[sources[b0 + i] + 1] == c->h_row_ptr[sources[b0 + i]]) active[i] = 0;
        std::vector<int32_t> nround((size_t)nb, 1);
        inactive.assign((size_t)nb, 0); // (see fora_hip_topk_batch: only the slots that never run a round need this copy)
        bool any_inactive = false;
        for (int i = 0; i < nb; i++) { inactive[i] = active[i] ? 0 : 1; any_inactive |= inactive[i] != 0; }
        if (any_inactive) {
            HIPCHK(c, hipMemcpyAsync(c->d_active, inactive.data(), (size_t)nb, hipMemcpyHostToDevice, c->stream));
            hipLaunchKernelGGL(k_copy_slab, dim3(chunks, nb), dim3(BLOCK), 0, c->stream, c->n, c->d_ppr, c->d_ppr2, (const uint8_t *)c->d_active);
        }
        double delta = init_delta;
        int round = 0;
        while (delta >= min_delta) { // query.h:944
            bool any = false;
            for (int i = 0; i < nb; i++) any |= active[i] != 0;
            if (!any) break;
            round++;
            double rmax = epsilon * sqrt(delta / 3 / m / L); // fora_setting with the round's delta, algo.h:455-463
            rmax *= rmax_scale;
            const double omega = (2 + epsilon) * L / delta / epsilon / epsilon;
            for (int i = 0; i < nb; i++) if (active[i]) nround[i] = round;
            HIPCHK(c, hipMemcpyAsync(c->d_active, active.data(), (size_t)nb, hipMemcpyHostToDevice, c->stream));
            HIPCHK(c, hipMemsetAsync(c->d_counters, 0, N_COUNTERS * sizeof(unsigned long long), c->stream));
            HIPCHK(c, hipMemsetAsync(c->d_above, 0, (size_t)nb * 8, c->stream));
            HIPCHK(c, hipMemsetAsync(c->d_fail, 0, (size_t)nb * 4, c->stream));
            HIPCHK(c, hipMemsetAsync(c->d_round_walks, 0, (size_t)nb * 8, c->stream));
            HIPCHK(c, hipMemsetAsync(c->d_wit_count, 0, (size_t)c->B * 4 * CSTRIDE, c->stream));
            rc = reset_binned_counters(c);
            if (rc) return rc;
            d = make_dev(c, nb, with_idx != 0, rmax, omega);
            int h = ev_begin(c, 4);
            hipLaunchKernelGGL(k_topk_frontier, dim3(chunks, nb), dim3(BLOCK), 0, c->stream, d, (const uint8_t *)c->d_active);
            ev_end(c, h);
            rc = run_push_levels(c, d, nullptr, 0, true); // algo.h:1020-1093
            if (rc) return rc;
            // compute_ppr_with_fwdidx_topk_with_bound, query.h:639-750, into ppr2
            h = ev_begin(c, 4);
            hipLaunchKernelGGL(k_copy_slab, dim3(chunks, nb), dim3(BLOCK), 0, c->stream, c->n, c->d_ppr, c->d_ppr2,
                               (const uint8_t *)c->d_active);
            ev_end(c, h);
            Dev dw = d;
            dw.ppr = c->d_ppr2;
            h = ev_begin(c, 2);
            hipLaunchKernelGGL(k_walk_alloc<ALLOC_BOUND>, (dw.wide && (dw.slot_major & 8u)) ? dim3(nb, chunks) : dim3(chunks, nb), dim3(BLOCK), 0, c->stream, dw, with_idx ? 1 : 0,
                               (const uint8_t *)c->d_active, c->d_cursor, c->d_round_walks, c->cursor_epoch);
            ev_end(c, h);
            launch_walks(c, dw, nb, with_idx != 0, (uint32_t)round, 0);
            h = ev_begin(c, 4);
            if (delta < threshold) // query.h:745-746
                hipLaunchKernelGGL(k_bounds_update, dim3(chunks, nb), dim3(BLOCK), 0, c->stream, dw, (const uint64_t *)c->d_ppr,
                                   (const uint8_t *)c->d_active, (const unsigned long long *)c->d_round_walks, L,
                                   1.0 / c->n, sqrt(1.0 / c->n), c->d_upper, c->d_lower);
            // if_stop, algo.h:1096-1166
            hipLaunchKernelGGL(k_count_above, dim3(std::min<uint32_t>(chunks, 256), nb), dim3(BLOCK), 0, c->stream, dw,
                               (const uint8_t *)c->d_active, 2.0 * delta, c->d_above);
            const bool bounds_on = !(delta >= threshold);
            if (bounds_on) {
                Dev dl = dw;
                dl.ppr = (uint64_t *)c->d_lower; // non-negative f64: bit patterns order like the values
                rc = launch_select(c, dl, nb, k, c->d_lb_ids, c->d_lb_sc, 1);
                if (rc) return rc;
                hipLaunchKernelGGL(k_bound_ratio, dim3(nb), dim3(SEL_THREADS), 0, c->stream, dw, k, (const int32_t *)c->d_lb_ids,
                                   (const double *)c->d_lb_sc, (const uint8_t *)c->d_active, (const double *)c->d_upper,
                                   1.0 + epsilon, c->d_filter, c->d_fail);
                hipLaunchKernelGGL(k_bound_scan, dim3(chunks, nb), dim3(BLOCK), 0, c->stream, dw, k, (const double *)c->d_lb_sc,
                                   (const uint8_t *)c->d_active, (const double *)c->d_upper, (const double *)c->d_lower, delta,
                                   1.0 + epsilon, (1 + epsilon) / (1 - epsilon), c->d_filter, c->d_fail);
            }
            ev_end(c, h);
            above.assign((size_t)nb, 0);
            failv.assign((size_t)nb, 0);
            HIPCHK(c, hipMemcpyAsync(above.data(), c->d_above, (size_t)nb * 8, hipMemcpyDeviceToHost, c->stream));
            HIPCHK(c, hipMemcpyAsync(failv.data(), c->d_fail, (size_t)nb * 4, hipMemcpyDeviceToHost, c->stream));
            rc = check_dev_err(c);
            if (rc) return rc;
            for (int i = 0; i < nb; i++) {
                if (!active[i]) continue;
                const bool stop = above[i] >= (unsigned long long)k || (bounds_on && failv[i] == 0);
                if (stop || delta <= min_delta) active[i] = 0; // query.h:962-964
            }
            if (delta <= min_delta) break;
            delta = std::max(min_delta, delta / 2.0); // query.h:966
        }
        Dev ds = make_dev(c, nb, false);
        ds.ppr = c->d_ppr2;
        int h = ev_begin(c, 4);
        rc = launch_select(c, ds, nb, k, c->d_topk_ids, c->d_topk_sc, 0);
        if (rc) return rc;
        ev_end(c, h);
        ev_end(c, hb);
        HIPCHK(c, hipMemcpyAsync(ids + (size_t)b0 * k, c->d_topk_ids, (size_t)nb * k * 4, hipMemcpyDeviceToHost, c->stream));
        HIPCHK(c, hipMemcpyAsync(scores + (size_t)b0 * k, c->d_topk_sc, (size_t)nb * k * 8, hipMemcpyDeviceToHost, c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));
        hipError_t e = hipGetLastError();
        if (e != hipSuccess) return fail(c, FORA_E_HIP, std::string("topk (bounds): ") + hipGetErrorString(e));
        ev_collect(c);
        HIPCHK(c, hipMemcpy(c->h_qs.data(), c->d_qs, (size_t)nb * sizeof(QState), hipMemcpyDeviceToHost));
        for (int i = 0; i < nb; i++) {
            c->timing.pops += c->h_qs[i].pops;
            c->timing.relax += c->h_qs[i].relax;
            c->timing.walks += c->h_qs[i].n_walks;
            c->timing.idx_hits += c->h_qs[i].n_hit;
        }
        if (rounds) for (int i = 0; i < nb; i++) rounds[b0 + i] = nround[i];
    }
    return FORA_OK;
}

// gen_exact_topk's kernel (query.h:1192-1238): the push with threshold = one unit per out-edge and a fixed
// number of levels; what it reserves is the exact PPR up to (1-alpha)^max_iter.
static int power_iteration_batch_impl(fora_ctx *c, const int32_t *sources, int nq, int max_iter, double *ppr_out,
                                      uint64_t *ppr_fix_out, int k, int32_t *ids, double *scores);
int fora_hip_power_iteration_batch(fora_ctx *c, const int32_t *sources, int nq, int max_iter, double *ppr_out,
                                   uint64_t *ppr_fix_out, int k, int32_t *ids, double *scores) {
    return with_bucket_retry(c, [&] { return power_iteration_batch_impl(c, sources, nq, max_iter, ppr_out, ppr_fix_out, k, ids, scores); });
}
static int power_iteration_batch_impl(fora_ctx *c, const int32_t *sources, int nq, int max_iter, double *ppr_out,
                                      uint64_t *ppr_fix_out, int k, int32_t *ids, double *scores) {
    if (!c) return FORA_E_ARG;
    if (!c->n) return fail(c, FORA_E_ARG, "set_graph first");
    if (!c->have_params) return fail(c, FORA_E_ARG, "set_params first (alpha)");
    if (nq < 0 || (nq && !sources) || max_iter < 1 || max_iter >= MAX_LEVELS) return fail(c, FORA_E_ARG, "bad arguments");
    const bool want_topk = ids || scores;
    if (want_topk && (!ids || !scores || k < 1 || k > SEL_MAXK || k > c->n)) return fail(c, FORA_E_ARG, "bad k / ids / scores");
    for (int i = 0; i < nq; i++)
        if (sources[i] < 0 || sources[i] >= c->n) return fail(c, FORA_E_ARG, "source id out of range");
    HIPCHK(c, hipSetDevice(c->device));
    c->bk_div = 1;
    int rc = ensure_workspace(c, nq, c->omega);
    if (rc) return rc;
    const uint64_t n = (uint64_t)c->n;
    if (want_topk && c->topk_cap < c->B * k) {
        dfree(c->d_topk_ids); dfree(c->d_topk_sc);
        HIPCHK(c, hipMalloc(&c->d_topk_ids, (size_t)c->B * k * 4));
        HIPCHK(c, hipMalloc(&c->d_topk_sc, (size_t)c->B * k * 8));
        c->topk_cap = c->B * k;
    }
    const int per = even_batch(nq, c->B);
    for (int b0 = 0; b0 < nq; b0 += per) {
        const int nb = std::min(per, nq - b0);
        const int hb = ev_begin(c, 5);
        rc = reset_batch_state(c, nb, sources + b0);
        if (rc) return rc;
        Dev d = make_dev(c, nb, false, 0.0, c->omega); // rmax 0 -> threshold of one unit per out-edge
        hipLaunchKernelGGL(k_init_batch, dim3((nb + BLOCK - 1) / BLOCK), dim3(BLOCK), 0, c->stream, d, 2);
        rc = run_push_levels(c, d, nullptr, max_iter);
        if (rc) return rc;
        if (want_topk) {
            int h = ev_begin(c, 4);
            rc = launch_select(c, d, nb, k, c->d_topk_ids, c->d_topk_sc, 0);
            if (rc) return rc;
            ev_end(c, h);
            HIPCHK(c, hipMemcpyAsync(ids + (size_t)b0 * k, c->d_topk_ids, (size_t)nb * k * 4, hipMemcpyDeviceToHost, c->stream));
            HIPCHK(c, hipMemcpyAsync(scores + (size_t)b0 * k, c->d_topk_sc, (size_t)nb * k * 8, hipMemcpyDeviceToHost, c->stream));
        }
        ev_end(c, hb);
        rc = check_dev_err(c);
        if (rc) return rc;
        hipError_t e = hipGetLastError();
        if (e != hipSuccess) return fail(c, FORA_E_HIP, std::string("power iteration: ") + hipGetErrorString(e));
        ev_collect(c);
        const uint64_t bytes = (uint64_t)nb * n * 8;
        if (ppr_fix_out) HIPCHK(c, hipMemcpy(ppr_fix_out + (uint64_t)b0 * n, c->d_ppr, bytes, hipMemcpyDeviceToHost));
        if (ppr_out) {
            double *dst = ppr_out + (uint64_t)b0 * n;
            HIPCHK(c, hipMemcpy(dst, c->d_ppr, bytes, hipMemcpyDeviceToHost));
            uint64_t *raw = (uint64_t *)dst;
            for (uint64_t i = 0; i < (uint64_t)nb * n; i++) dst[i] = std::ldexp((double)raw[i], -62);
        }
    }
    return FORA_OK;
}

int fora_hip_reset_timing(fora_ctx *c) {
    if (!c) return FORA_E_ARG;
    c->timing = fora_timing{};
    if (c->twin) c->twin->timing = fora_timing{};
    (void)hipSetDevice(c->device);
    (void)hipMemset(c->d_stamps, 0, 32 * sizeof(unsigned long long));
    return FORA_OK;
}
int fora_hip_get_stamps(fora_ctx *c, uint64_t *out32) {
    if (!c || !out32) return FORA_E_ARG;
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipMemcpy(out32, c->d_stamps, 32 * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    return FORA_OK;
}
int fora_hip_get_timing(fora_ctx *c, fora_timing *out) {
    if (!c || !out) return FORA_E_ARG;
    *out = c->timing;
    if (c->twin) {
        const fora_timing &w = c->twin->timing;
        out->push_pop_ms += w.push_pop_ms; out->push_expand_ms += w.push_expand_ms; out->push_accum_ms += w.push_accum_ms;
        out->walk_alloc_ms += w.walk_alloc_ms; out->walk_ms += w.walk_ms; out->walk_accum_ms += w.walk_accum_ms;
        out->other_ms += w.other_ms; out->batch_ms += w.batch_ms;
        out->push_pop_launches += w.push_pop_launches; out->push_expand_launches += w.push_expand_launches;
        out->push_accum_launches += w.push_accum_launches; out->walk_launches += w.walk_launches; out->batches += w.batches;
        out->pops += w.pops; out->relax += w.relax; out->walks += w.walks; out->walk_steps += w.walk_steps; out->levels += w.levels;
        out->idx_hits += w.idx_hits;
        out->push_tail_ms += w.push_tail_ms; out->push_tail_launches += w.push_tail_launches;
    }
    return FORA_OK;
}

} // extern "C"
