// bk_tree.cpp -- native PUCT search core and lock-step game pool (host only, part of libbkgo.so).
//
// Same search as the reference's bokego/mcts.py:110-234 (and as bokego_amd/mcts.py, against which
// tests compare it visit for visit): PUCT selection  -avg + c*P(move)*sqrt(sum N)/(1+N), a leaf is
// expanded on the visit after N > expand_thresh, the value of the last node of the path is backed
// up with alternating sign, the most visited child is chosen and becomes the new root with its
// subtree kept.  Positions equal in (board, ko, last move, side) share one node (the reference's
// dicts are keyed that way, mcts.py:294-299).
//
// A pool advances many independent games; each game runs until it needs network outputs, the
// pool gathers every game's request into ONE batch of feature planes (policy positions first),
// the caller evaluates it on the GPU and hands the results back.  Nothing here touches HIP.
#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <condition_variable>
#include <cstdlib>
#include <cstring>
#include <exception>
#include <functional>
#include <mutex>
#include <thread>
#include <pthread.h>
#include <unordered_map>
#include <vector>

#include "../../include/bokego_go.h"
#include "../../include/bokego_tree.h"

namespace {

// 48 bytes: the search (select / backprop) walks these and nothing else; the 192-byte position of node i is poses[i]
// (read when a node is interned, expanded or sent for evaluation).  A node's children are interned together, so their
// statistics are mostly neighbours in memory.  Measured on a synthetic 72-child select (tools/micro/select_bench.cpp):
// 232-byte nodes with two divisions per child 350 ns, these nodes 240 ns, + the cached average 185 ns.
struct TNode {
    double V = 0.0;
    double avg = 0.0;  // N == 0 ? 0 : V / N, refreshed where V and N change (backprop): select divides once per child
                       // (simulation mode: ((1 - w) Q + w V) / N, mcts.py:227-230)
    int N = 0;
    float value = 0.f;
    uint8_t has_value = 0, has_prior = 0, expanded = 0, terminal = 0;
    uint8_t speculated = 0;  // its policy and its children's values were requested ahead of its expansion
    int kids_off = 0, n_kids = 0;
    int prior_off = -1;
    int mv = BK_NO_MOVE;
};

constexpr int kStepTrim = 4;   // children an expansion leaves to later requests to stay within a step of prm.request_steps
enum State { S_INIT, S_ROOT_EXPAND, S_WAIT_ROOT, S_ROOT_READY, S_SEARCH, S_PLAYOUT, S_WAIT_LEAF, S_CHOOSE, S_DONE, S_IDLE };

struct Rng {  // xoshiro256** seeded by splitmix64: per-game stream, independent of how games are sharded
    uint64_t s[4];
    explicit Rng(uint64_t seed) {
        uint64_t x = seed;
        for (auto& v : s) {
            uint64_t z = (x += 0x9E3779B97F4A7C15ull);
            z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
            z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
            v = z ^ (z >> 31);
        }
    }
    static uint64_t rotl(uint64_t x, int k) { return (x << k) | (x >> (64 - k)); }
    uint64_t next() {
        const uint64_t r = rotl(s[1] * 5, 7) * 9, t = s[1] << 17;
        s[2] ^= s[0]; s[3] ^= s[1]; s[1] ^= s[2]; s[0] ^= s[3]; s[2] ^= t; s[3] = rotl(s[3], 45);
        return r;
    }
    double uniform() { return (double)(next() >> 11) * (1.0 / 9007199254740992.0); }
    double normal() {
        double u, v, q;
        do { u = 2 * uniform() - 1; v = 2 * uniform() - 1; q = u * u + v * v; } while (q >= 1 || q == 0);
        return u * std::sqrt(-2 * std::log(q) / q);
    }
    double gamma(double a) {  // Marsaglia-Tsang; a < 1 via the U^(1/a) boost
        if (a < 1) return gamma(a + 1) * std::pow(uniform(), 1.0 / a);
        const double d = a - 1.0 / 3, c = 1 / std::sqrt(9 * d);
        for (;;) {
            double x, v;
            do { x = normal(); v = 1 + c * x; } while (v <= 0);
            v = v * v * v;
            const double u = uniform();
            if (u < 1 - 0.0331 * x * x * x * x || std::log(u) < 0.5 * x * x + d * (1 - v + std::log(v))) return d * v;
        }
    }
};

struct Game {
    bk_search_params prm;
    std::vector<TNode> nodes;
    std::vector<bk_pos> poses;          // parallel to nodes
    std::vector<double> Qs;             // parallel to nodes when prm.simulate: summed playout rewards (MCTS.Q[node]); else empty
    std::vector<int> kid_ids;
    std::vector<double> priors;
    // position -> node id: open addressing over a flat array (power-of-two size, linear probing, -1 = empty).  No per-bucket
    // allocations: with hundreds of games advancing on a team of threads, the allocator -- a game's memory is freed by
    // whichever thread advances it next -- was what kept 16 threads from being faster than 4 (profiles/r03_host_tree.txt).
    std::vector<int> slots;
    // scratch of prune() / expand() / want_best(), kept for their capacity
    std::vector<int> sc_remap, sc_order, sc_kids, sc_ids, sc_w;
    std::vector<TNode> sc_nodes;
    std::vector<bk_pos> sc_poses;
    std::vector<double> sc_priors, sc_Q;
    int root = -1;
    State state = S_INIT;
    int remaining = 0;
    std::vector<int> path;
    // bk_search_params.leaves > 1 (opt-in, not the reference's search): the rollouts of this step that wait for their leaf's value, each
    // as [length, ids ...] back to back, in the order they were made; every node below the root on such a path carries a virtual loss
    std::vector<int> pend;
    int n_pend = 0;
    std::vector<int> req_policy, req_value;
    std::vector<int> spec_queue;        // leaves that reached prm.speculate visits and wait for a request with room
    std::vector<int> spill;             // children of expanded nodes whose value did not fit under prm.request_tasks: they
                                        // travel with the next requests that have room (or are asked for when selected)
    std::unordered_map<int, std::vector<int>> spec_kids;   // their would-be children's node ids
    Rng rng;
    std::vector<int16_t> moves;
    std::vector<std::vector<std::pair<int16_t, int32_t>>> visit_log;  // per ply: (move, N) of the root's children
    uint64_t n_value_evals = 0, n_policy_evals = 0, n_requests = 0;
    float final_score = 0.f;
    // visit / value statistics of the moves chosen so far (bk_pool_game_stats; the end-of-generation all-reduce sums them):
    // root_visits[m] = sum over the plies played of the visit count of the root's child for move m when the ply was chosen;
    // root values = V[root] / N[root] at that moment (what MCTS.winrate reads, mcts.py:159-170: 2 winrate - 1, side to move),
    // summed in 2^-32 fixed point so that the totals do not depend on the order games are added up in
    uint64_t root_visits[81] = {0};
    int64_t sum_root_value_q = 0, sum_abs_root_value_q = 0;
    uint64_t n_root_values = 0;
    bool manual = false;  // driven from outside: rollouts are added, moves chosen/played by the caller
    // MCTS.rollout(n, analyze_dict) (mcts.py:143-147): while on, every descent longer than two nodes is remembered
    // under the root child it went through (path[1] -> path[1:]; a later descent through the same child replaces it)
    bool analyze = false;
    std::unordered_map<int, std::vector<int>> variations;

    Game(const bk_search_params& p, uint64_t seed) : prm(p), rng(seed) {}

    static uint64_t key_hash(const bk_pos& p) { return p.hash ^ (0x9E3779B97F4A7C15ull * (uint64_t)(uint16_t)p.last_move); }
    static bool same(const bk_pos& a, const bk_pos& b) {
        return a.ko == b.ko && a.last_move == b.last_move && ((a.turn ^ b.turn) & 1) == 0 &&
               std::memcmp(a.board, b.board, 81) == 0;
    }
    static size_t mix(uint64_t h) { return (size_t)((h ^ (h >> 29)) * 0x9E3779B97F4A7C15ull >> 17); }
    int find(const bk_pos& p) const {   // node id of a position, -1 if the tree has never seen it
        if (slots.empty()) return -1;
        const size_t mask = slots.size() - 1;
        for (size_t i = mix(key_hash(p)) & mask;; i = (i + 1) & mask) {
            const int id = slots[i];
            if (id < 0) return -1;
            if (same(poses[id], p)) return id;
        }
    }
    void table_insert(int id) {
        const size_t mask = slots.size() - 1;
        size_t i = mix(key_hash(poses[id])) & mask;
        while (slots[i] >= 0) i = (i + 1) & mask;
        slots[i] = id;
    }
    void table_rebuild(size_t min_nodes) {   // every node re-entered into a table at most half full
        size_t cap = 1024;
        while (cap < 2 * min_nodes) cap <<= 1;
        slots.assign(cap, -1);
        for (int i = 0; i < (int)nodes.size(); ++i) table_insert(i);
    }
    int intern(const bk_pos& p) {
        const int have = find(p);
        if (have >= 0) return have;
        if (2 * (nodes.size() + 1) > slots.size()) table_rebuild(2 * (nodes.size() + 1));
        TNode n;
        n.mv = p.last_move;
        n.terminal = (p.turn > prm.max_turns || p.last_move == BK_PASS) ? 1 : 0;  // mcts.py:362-364
        nodes.push_back(n);
        poses.push_back(p);
        if (prm.simulate) Qs.push_back(0.0);
        table_insert((int)nodes.size() - 1);
        return (int)nodes.size() - 1;
    }

    // ---- request size -----------------------------------------------------------------------------------------------------
    // A request of P policy rows and V value-only rows is 2 P + V network tasks for the engine (a policy row runs both nets).
    // prm.request_tasks > 0 keeps a request within that many tasks where the search allows it (the fp32 engine gives a board 4
    // CUs up to 64 tasks and 3 from 65: 104 -> 140 us): values that are wanted but not needed yet wait in `spill` for a later
    // request.  The networks are pure functions, so WHEN a value is computed never changes the search.
    int request_tasks() const { return 2 * (int)req_policy.size() + (int)req_value.size(); }
    // the size the request may grow to at no extra cost: the first step of prm.request_steps that holds what is in it already
    // (an expansion with unknown priors may have put more than the first step in: it still takes passengers up to ITS step)
    int ceiling() const {
        const int cur = std::max(1, request_tasks());
        for (int st : prm.request_steps)
            if (st > 0 && cur <= st) return st;
        return (cur + 255) / 256 * 256;   // beyond the last step: whole rounds of one-CU workgroups
    }
    bool fits(int extra_tasks) const { return prm.request_tasks <= 0 || request_tasks() + extra_tasks <= ceiling(); }
    // rows one collect can take (the pool's smallest `cap`): passengers -- spilled values, speculative rows -- never push a
    // request beyond it; a request that cannot fit would be skipped by every collect and its game would stall
    int row_cap = 1 << 30;
    bool room_for_rows(int extra_rows) const { return (int)(req_policy.size() + req_value.size()) + extra_rows <= row_cap; }
    bool queued_value(int c) const { return std::find(req_value.begin(), req_value.end(), c) != req_value.end(); }
    // children (node ids) that still lack a value and are not in the request, most promising first when the parent's priors are
    // known (the search visits high priors first), else in move order
    std::vector<int> wanting(const std::vector<int>& kids, int parent) const {
        std::vector<int> w;
        for (int c : kids)
            if (!nodes[c].has_value && !queued_value(c)) w.push_back(c);
        if (prm.request_tasks > 0 && nodes[parent].has_prior) {
            const double* pr = &priors[nodes[parent].prior_off];
            std::stable_sort(w.begin(), w.end(), [&](int x, int y) { return pr[nodes[x].mv] > pr[nodes[y].mv]; });
        }
        return w;
    }

    // branch_num (mcts.py:189-190, 309-317): the node's children are the legal moves among the k best-prior moves
    bool branching() const { return prm.branch_num > 0 && prm.branch_num < 81; }
    int pending_expand = -1;      // a node whose expansion waits for its policy (branch_num)

    // mcts.py:185-192 + the eager evaluation of the new children
    void expand(int id) {
        if (nodes[id].expanded) return;
        if (branching() && !nodes[id].terminal && !nodes[id].has_prior) {
            // which moves become children depends on the priors: ask for them, expand when they are back (finish_expand)
            auto queued = std::find(req_policy.begin(), req_policy.end(), id) != req_policy.end();
            if (!queued) req_policy.push_back(id);
            pending_expand = id;
            return;
        }
        bk_pos kids[81];
        int16_t mv[81];
        int n = 0;
        if (!nodes[id].terminal) n = bk_pos_children(&poses[id], kids, mv);
        if (branching() && n > 0) {
            const double* pr = &priors[nodes[id].prior_off];
            int order[81];
            for (int i = 0; i < 81; ++i) order[i] = i;
            std::stable_sort(order, order + 81, [&](int a, int b) { return pr[a] > pr[b]; });
            bool top[81] = {false};
            for (int i = 0; i < prm.branch_num; ++i) top[order[i]] = true;
            int m = 0;
            for (int i = 0; i < n; ++i)
                if (mv[i] >= 0 && mv[i] < 81 && top[mv[i]]) { kids[m] = kids[i]; mv[m] = mv[i]; ++m; }
            n = m;
        }
        const int off = (int)kid_ids.size();
        for (int i = 0; i < n; ++i) kid_ids.push_back(intern(kids[i]));  // may reallocate `nodes`
        TNode& nd = nodes[id];
        nd.kids_off = off;
        nd.n_kids = n;
        nd.expanded = 1;
        if (!nd.has_prior) req_policy.push_back(id);
        sc_ids.assign(kid_ids.begin() + off, kid_ids.begin() + off + n);
        const std::vector<int>& kidv = sc_ids;
        if (prm.eager_top > 0) {
            // only the children the search is going to visit: it visits a node's unvisited children in the order of their
            // priors (PUCT with N = 0, avg = 0) and, measured over 1600-rollout searches, ever visits 4 of ~75 in the median
            // (10 at the 90th percentile).  With known priors the best eager_top go now; with unknown ones (the policy row is
            // in this request) nothing does, and the first descent into the node asks for what it needs (request_leaf).
            if (nodes[id].has_prior) want_best(kidv, id, prm.eager_top);
        } else if (prm.eager) {
            // with unknown priors (the policy row travels in this very request) nothing says which child the search wants
            // first: every child goes now.  With known priors (the node was evaluated ahead) the tail may wait.
            const bool may_wait = prm.request_tasks > 0 && nodes[id].has_prior;
            const std::vector<int> w = wanting(kidv, id);
            // An expansion with unknown priors that comes out just over a step of the request sizes -- 79 children + the policy
            // row = 81 tasks, where the engine's launch gives a board 3 CUs up to 80 tasks and 2 from 81 (135 -> 174 us) -- leaves
            // its last children (move order: nothing says which matter) to later requests, like the overflow of any other
            // expansion; one of them is needed before it has travelled in about one case in fifty.
            size_t keep = w.size();
            if (!may_wait && prm.request_tasks > 0) {
                const int need = request_tasks() + (int)w.size();
                for (int st : prm.request_steps)
                    if (st > 0 && need > st && need - st <= kStepTrim && need - st < (int)w.size()) keep = w.size() - (size_t)(need - st);
            }
            for (size_t i = 0; i < w.size(); ++i) {
                const int c = w[i];
                if (i < keep && (!may_wait || (fits(1) && room_for_rows(1)))) req_value.push_back(c);
                else spill.push_back(c);
            }
        }
    }

    // request the values of the `k` best-prior children of `parent` that still lack one (parent's priors are known)
    void want_best(const std::vector<int>& kids, int parent, int k) {
        const double* pr = &priors[nodes[parent].prior_off];
        std::vector<int>& w = sc_w;
        w.clear();
        for (int c : kids)
            if (!nodes[c].has_value && !queued_value(c)) w.push_back(c);
        if ((int)w.size() > k) {
            std::partial_sort(w.begin(), w.begin() + k, w.end(), [&](int x, int y) {
                return pr[nodes[x].mv] > pr[nodes[y].mv] || (pr[nodes[x].mv] == pr[nodes[y].mv] && nodes[x].mv < nodes[y].mv);
            });
            w.resize(k);
        }
        for (int c : w) req_value.push_back(c);
    }

    // the rollout's last node needs its value for the backup (mcts.py:151: leaf.value, evaluated on first use).  With
    // eager_top the request also takes the next-best unevaluated siblings along: where the search got this far down a
    // node's prior order it tends to go on (all visited children looking bad), and one request then serves the next visits.
    void request_leaf() {
        const int leaf = path.back();
        if (!prm.use_value || nodes[leaf].has_value || queued_value(leaf)) return;
        req_value.push_back(leaf);
        if (prm.eager_top > 0 && path.size() >= 2) {
            const int parent = path[path.size() - 2];
            const TNode& pn = nodes[parent];
            if (pn.has_prior && pn.n_kids > 0) {
                sc_ids.assign(kid_ids.begin() + pn.kids_off, kid_ids.begin() + pn.kids_off + pn.n_kids);
                want_best(sc_ids, parent, prm.eager_top - 1);
            }
        }
    }

    int select(int id) const {  // mcts.py:219-234
        const TNode& nd = nodes[id];
        const int* kids = &kid_ids[nd.kids_off];
        long total = 0;
        for (int i = 0; i < nd.n_kids; ++i) total += nodes[kids[i]].N;
        if (total == 0) total = 1;
        const double sq = std::sqrt((double)total), c = prm.c_puct;
        const double* prior = &priors[nd.prior_off];
        int best = -1;
        double best_s = 0;
        for (int i = 0; i < nd.n_kids; ++i) {
            const TNode& k = nodes[kids[i]];
            // an unvisited child (N = 0, avg = +0.0) scores -0.0 + e / 1.0 = e exactly: no division for ~70 of a node's ~75 children
            const double e = c * prior[k.mv] * sq;
            const double s = k.N == 0 ? e : -k.avg + (e / (double)(1 + k.N));
            if (best < 0 || s > best_s) { best = kids[i]; best_s = s; }
        }
        return best;
    }

    // Evaluate ahead (prm.speculate): request id's policy and the values of the children it would get, WITHOUT expanding
    // it -- the children are interned (unlinked nodes) so that the later expand() finds them evaluated.
    bool speculate(int id) {  // false: not (completely) in this request yet: stays a candidate
        if (nodes[id].expanded || nodes[id].terminal || nodes[id].speculated) return true;
        std::vector<int>& cid = spec_kids[id];   // the would-be children, interned once (a candidate may be retried)
        if (cid.empty()) {
            bk_pos kids[81];
            int16_t mv[81];
            const int n = bk_pos_children(&poses[id], kids, mv);
            for (int i = 0; i < n; ++i) cid.push_back(intern(kids[i]));  // may reallocate `nodes`
        }
        auto queued = [](const std::vector<int>& v, int x) { return std::find(v.begin(), v.end(), x) != v.end(); };
        const bool want_p = !nodes[id].has_prior && !queued(req_policy, id);
        if (prm.eager_top > 0) {
            // staged: the policy row goes first (2 tasks: it fits any request); once the priors are back, the eager_top
            // best children -- exactly what expand() would ask for, so that the expansion needs no request of its own
            if (!nodes[id].has_prior) {
                if (want_p && (int)(req_policy.size() + req_value.size()) + 1 <= prm.speculate_rows && fits(2)) req_policy.push_back(id);
                return false;
            }
            const size_t before = req_value.size();
            if ((int)(req_policy.size() + before) + prm.eager_top > prm.speculate_rows || !fits(prm.eager_top)) return false;
            want_best(cid, id, prm.eager_top);
            nodes[id].speculated = 1;
            spec_kids.erase(id);
            return true;
        }
        const std::vector<int> w = wanting(cid, id);
        const int rows = (want_p ? 1 : 0) + (int)w.size();
        auto room = [&](int more_rows, int more_tasks) {
            return (int)(req_policy.size() + req_value.size()) + more_rows <= prm.speculate_rows && fits(more_tasks);
        };
        if (prm.request_tasks <= 0) {   // all or nothing: the candidate goes when it fits as a whole
            if (!room(rows, rows + (want_p ? 1 : 0))) return false;
        }
        if (want_p) {
            if (!room(1, 2)) return false;
            req_policy.push_back(id);
        }
        size_t k = 0;
        for (; k < w.size() && room(1, 1); ++k) req_value.push_back(w[k]);
        if (k < w.size()) return false;          // the rest (by prior, once the policy row is back) with a later request
        nodes[id].speculated = 1;
        spec_kids.erase(id);
        return true;
    }
    void add_speculation() {  // a request is going out anyway: let it carry what is waiting, then the candidates that fit
        if (!spill.empty()) {
            std::vector<int> left;
            for (int c : spill) {
                if (nodes[c].has_value || queued_value(c)) continue;
                if (fits(1) && room_for_rows(1)) req_value.push_back(c);
                else left.push_back(c);
            }
            spill.swap(left);
        }
        std::vector<int> keep;
        for (int id : spec_queue)
            if (!nodes[id].expanded && !nodes[id].speculated && !nodes[id].terminal) keep.push_back(id);
        std::stable_sort(keep.begin(), keep.end(), [&](int x, int y) { return nodes[x].N > nodes[y].N; });
        if (keep.size() > 8) {   // candidates that never fit are retried with every request: keep the list short
            for (size_t i = 8; i < keep.size(); ++i) spec_kids.erase(keep[i]);
            keep.resize(8);
        }
        spec_queue.clear();
        for (int id : keep)
            if (!speculate(id)) spec_queue.push_back(id);   // does not fit this time: stays a candidate
        if (spec_queue.empty()) spec_kids.clear();
    }

    // ---- simulation mode (MCTS(no_sim=False), boke.py --simulate; mcts.py:147-148,195-206) ---------------------------------
    // A rollout's score is the result of a playout from its leaf: moves sampled from the policy (Go_MCTS.get_move,
    // mcts.py:348-360: a sampled move that is illegal or fills the mover's own one-point eye is zeroed in the node's
    // distribution and another one drawn), to the end of the game (a pass or turn > max_turns, mcts.py:362-364), scored by
    // Game.score (mcts.py:338; the gnugo scorer needs a binary).  Where no acceptable move is left the playout PASSES -- what
    // the reference's docstring says; its own code raises there (the branch needs 82 non-zero probabilities) and no search
    // of it gets through its first rollouts (tests/golden/simulate_playouts.json).
    // A playout position that is a node of the tree (the leaf itself, a transposition) uses -- and changes, as the
    // reference's cached distribution is changed -- that node's priors; any other one lives only for the playout: nodes with
    // an id >= po_mark keep their priors in po_priors and are dropped when the playout ends (the reference keeps every
    // position it ever evaluated, without bound).  Draws come from the game's own generator (inverse CDF over the
    // unnormalised probabilities), not from torch's.
    int po = -1, po_mark = -1, po_reward = 0;
    std::vector<double> po_priors;
    bool transient(int id) const { return po_mark >= 0 && id >= po_mark; }
    double* prior_ptr(int id) { return transient(id) ? &po_priors[nodes[id].prior_off] : &priors[nodes[id].prior_off]; }

    int sample_move(int id) {
        double* pr = prior_ptr(id);
        const bk_pos& ps = poses[id];
        const int color = (ps.turn & 1) ? BK_WHITE : BK_BLACK;
        auto draw = [&]() {
            double tot = 0;
            for (int i = 0; i < 81; ++i) tot += pr[i];
            if (!(tot > 0)) return -1;
            const double u = rng.uniform() * tot;
            double c = 0;
            int last = -1;
            for (int i = 0; i < 81; ++i) {
                if (!(pr[i] > 0)) continue;
                c += pr[i];
                last = i;
                if (u < c) return i;
            }
            return last;
        };
        int mv = draw();
        for (int tries = 0; mv >= 0 && (!bk_pos_is_legal(&ps, mv) || bk_pos_possible_eye(&ps, mv) == color); ++tries) {
            if (tries >= 81) return BK_PASS;
            pr[mv] = 0;
            mv = draw();
        }
        return mv < 0 ? BK_PASS : mv;
    }

    void drop_transient() {   // the playout's own nodes, newest first: with linear probing the newest entry is always safe to remove
        if (po_mark < 0) return;
        const size_t mask = slots.size() - 1;
        for (int id = (int)nodes.size() - 1; id >= po_mark; --id) {
            size_t i = mix(key_hash(poses[id])) & mask;
            while (slots[i] != id) i = (i + 1) & mask;
            slots[i] = -1;
        }
        nodes.resize((size_t)po_mark);
        poses.resize((size_t)po_mark);
        Qs.resize((size_t)po_mark);
        po_priors.clear();
        po_mark = -1;
    }

    // true: the playout is over (po_reward is the rollout's score, from the point of view of the side to move at the leaf,
    // mcts.py:199-204); false: a position's policy has been asked for
    bool playout() {
        if (po < 0) {
            po = path.back();
            po_mark = (int)nodes.size();
        }
        while (!nodes[po].terminal) {
            if (!nodes[po].has_prior) {
                if (std::find(req_policy.begin(), req_policy.end(), po) == req_policy.end()) req_policy.push_back(po);
                return false;
            }
            const int mv = sample_move(po);
            bk_pos c = poses[po];
            bk_pos_play(&c, mv);
            po = intern(c);
        }
        int r = bk_pos_score(&poses[po], prm.komi) > 0 ? 1 : -1;
        if (poses[path.back()].turn & 1) r = -r;
        po_reward = r;
        po = -1;
        drop_transient();
        return true;
    }

    void backprop() {  // mcts.py:208-217
        if (prm.simulate || !prm.use_value || prm.value_weight != 1.0) {
            double v = (double)nodes[path.back()].value, r = (double)po_reward;
            const double w = prm.value_weight;
            for (int i = (int)path.size() - 1; i >= 0; --i) {
                TNode& n = nodes[path[i]];
                double q = 0.0;
                n.N += 1;
                if (prm.simulate) { q = (Qs[path[i]] += r); r = -r; }
                if (prm.use_value) { n.V += v; v = -v; }
                n.avg = ((1 - w) * q + w * n.V) / (double)n.N;
            }
            return;
        }
        if (prm.speculate > 0) {
            const TNode& leaf = nodes[path.back()];
            if (!leaf.expanded && !leaf.terminal && !leaf.speculated && leaf.N + 1 == prm.speculate) spec_queue.push_back(path.back());
        }
        double v = (double)nodes[path.back()].value;
        for (int i = (int)path.size() - 1; i >= 0; --i) {
            TNode& n = nodes[path[i]];
            n.N += 1;
            n.V += v;
            n.avg = n.V / (double)n.N;
            v = -v;
        }
    }

    void add_noise(int id) {  // mcts.py:366-369 with a per-game generator
        if (prm.noise_weight <= 0.f || !nodes[id].has_prior) return;   // (a restored game whose root has no priors yet: nothing to mix into)
        double g[81], sum = 0;
        for (auto& x : g) { x = rng.gamma(0.1); sum += x; }
        double* p = &priors[nodes[id].prior_off];
        const float w = prm.noise_weight;
        for (int i = 0; i < 81; ++i) p[i] = (double)((1.f - w) * (float)p[i] + w * (float)(g[i] / sum));
    }

    void note_choice() {  // the root's statistics at the moment a move is chosen (see root_visits)
        const TNode& r = nodes[root];
        for (int i = 0; i < r.n_kids; ++i) {
            const TNode& k = nodes[kid_ids[r.kids_off + i]];
            if (k.mv >= 0 && k.mv < 81) root_visits[k.mv] += (uint64_t)k.N;
        }
        if (r.N > 0) {
            const int64_t q = (int64_t)std::llround(r.V / (double)r.N * 4294967296.0);
            sum_root_value_q += q;
            sum_abs_root_value_q += q < 0 ? -q : q;
            n_root_values += 1;
        }
    }

    int pick_move() {  // most visited child, lowest move index on ties (mcts.py:122-128); early plies sampled
        const TNode& r = nodes[root];
        const int* kids = &kid_ids[r.kids_off];
        if ((int)moves.size() < prm.sample_plies) {
            long tot = 0;
            for (int i = 0; i < r.n_kids; ++i) tot += nodes[kids[i]].N;
            if (tot > 0) {
                long t = (long)(rng.uniform() * (double)tot);
                for (int i = 0; i < r.n_kids; ++i) {
                    t -= nodes[kids[i]].N;
                    if (t < 0) return kids[i];
                }
            }
        }
        int best = -1, best_n = -1;
        for (int i = 0; i < r.n_kids; ++i)
            if (nodes[kids[i]].N > best_n) { best = kids[i]; best_n = nodes[kids[i]].N; }
        return best;
    }

    void prune() {  // keep only the new root's subtree (bounds memory over a whole game); no allocation once warm
        std::vector<int>& remap = sc_remap;
        std::vector<int>& order = sc_order;
        remap.assign(nodes.size(), -1);
        order.clear();
        order.push_back(root);
        remap[root] = 0;
        for (size_t i = 0; i < order.size(); ++i) {
            const TNode& n = nodes[order[i]];
            for (int k = 0; k < n.n_kids; ++k) {
                const int c = kid_ids[n.kids_off + k];
                if (remap[c] < 0) { remap[c] = (int)order.size(); order.push_back(c); }
            }
        }
        std::vector<TNode>& nn = sc_nodes;
        std::vector<bk_pos>& npos = sc_poses;
        std::vector<int>& nk = sc_kids;
        std::vector<double>& np = sc_priors;
        std::vector<double>& nq = sc_Q;
        nn.clear();
        npos.clear();
        nk.clear();
        np.clear();
        nq.clear();
        for (int old : order) {
            TNode n = nodes[old];
            const int off = (int)nk.size();
            for (int k = 0; k < n.n_kids; ++k) nk.push_back(remap[kid_ids[n.kids_off + k]]);
            n.kids_off = off;
            if (n.has_prior) {
                const int po = (int)np.size();
                np.insert(np.end(), priors.begin() + n.prior_off, priors.begin() + n.prior_off + 81);
                n.prior_off = po;
            }
            nn.push_back(n);
            npos.push_back(poses[old]);
            if (prm.simulate) nq.push_back(Qs[old]);
        }
        Qs.swap(nq);
        nodes.swap(nn);       // the old arrays become next time's scratch
        poses.swap(npos);
        kid_ids.swap(nk);
        priors.swap(np);
        size_t w = 0;
        for (int id : spec_queue)
            if (remap[id] >= 0) spec_queue[w++] = remap[id];
        spec_queue.resize(w);
        spec_kids.clear();
        w = 0;
        for (int id : spill)
            if (remap[id] >= 0) spill[w++] = remap[id];
        spill.resize(w);
        table_rebuild(nodes.size());
        root = 0;
    }

    bool has_request() const { return !req_policy.empty() || !req_value.empty(); }

    // ---- the opt-in multi-leaf mode (bk_search_params.leaves > 1; SURVEY 7.6: "virtual loss only as an opt-in throughput mode") ----
    // NOT the reference's search: up to `leaves` rollouts of one step wait for a value together.  A waiting rollout leaves a virtual
    // loss on every node of its path below the root -- one more visit, lost by the side that moved there (V + 1: select() scores a
    // child by -avg) -- so that the next descent of the step goes elsewhere; back_up_leaves() takes the losses off and backs the
    // rollouts up in the order they were made.  A rollout whose leaf value is known is backed up at once, unless it ends on a node
    // whose expansion still waits for its policy row (kids without priors: nothing can be selected below it, and backing it up at
    // once would send the step's remaining rollouts down the same path): that one waits too, as in the one-leaf search.
    void virtual_loss(TNode& n, int sign) const {
        n.N += sign;
        if (!prm.leaves_visit_only) n.V += (double)sign;     // (visit only: the waiting rollout counts as a visit, not as a loss)
        n.avg = n.N > 0 ? n.V / (double)n.N : 0.0;
    }
    bool search_leaves() {  // true: a request goes out (n_pend rollouts wait for it); false: this move's rollouts are done
        while (remaining > n_pend) {
            path.clear();
            int id = root;
            path.push_back(id);
            for (;;) {
                if (nodes[id].n_kids == 0) {
                    if (!nodes[id].expanded && nodes[id].N > prm.expand_thresh) expand(id);   // (may reallocate `nodes`)
                    break;
                }
                if (!nodes[id].has_prior) break;                 // its policy row is still out
                id = select(id);
                path.push_back(id);
            }
            if (analyze && path.size() > 2) variations[path[1]].assign(path.begin() + 1, path.end());
            request_leaf();
            const TNode& leaf = nodes[path.back()];
            if (leaf.has_value && !(leaf.n_kids > 0 && !leaf.has_prior)) {
                backprop();
                --remaining;
                continue;
            }
            for (size_t i = 1; i < path.size(); ++i) virtual_loss(nodes[path[i]], +1);
            pend.push_back((int)path.size());
            pend.insert(pend.end(), path.begin(), path.end());
            if (++n_pend >= prm.leaves) break;
        }
        path.clear();
        if (n_pend > 0 || has_request()) {
            add_speculation();                                   // (what spilled over from earlier expansions rides along)
            return true;
        }
        return false;
    }
    void back_up_leaves() {
        size_t at = 0;
        for (int r = 0; r < n_pend; ++r) {
            const int len = pend[at++];
            path.assign(pend.begin() + (long)at, pend.begin() + (long)at + len);
            at += (size_t)len;
            for (size_t i = 1; i < path.size(); ++i) virtual_loss(nodes[path[i]], -1);
            backprop();
            --remaining;
        }
        pend.clear();
        n_pend = 0;
        path.clear();
    }

    void reroot(int id) {  // MCTS.set_root (mcts.py:153-157): keep the subtree, expand the new root
        variations.clear();
        root = id;
        if (prm.prune) prune();
        state = S_ROOT_EXPAND;
    }

    // run until network outputs are needed (returns true) or the game is over (false)
    bool advance() {
        for (;;) {
            switch (state) {
                case S_INIT: {
                    if (root < 0) {
                        bk_pos p;
                        bk_pos_init(&p);
                        root = intern(p);
                    }
                    state = S_ROOT_EXPAND;
                    break;
                }
                case S_ROOT_EXPAND:
                    expand(root);
                    if (!nodes[root].has_prior && req_policy.empty()) req_policy.push_back(root);
                    state = S_WAIT_ROOT;
                    if (has_request()) {
                        add_speculation();
                        return true;
                    }
                    break;
                case S_WAIT_ROOT:
                    if (pending_expand >= 0) {                       // branch_num: the root's priors are in
                        const int id = pending_expand;
                        pending_expand = -1;
                        expand(id);
                        if (has_request()) {                         // its best children's values (eager)
                            add_speculation();
                            return true;
                        }
                    }
                    state = S_ROOT_READY;
                    break;
                case S_ROOT_READY:
                    add_noise(root);
                    if (manual) {
                        state = remaining > 0 ? S_SEARCH : S_IDLE;
                        break;
                    }
                    if (nodes[root].terminal || nodes[root].n_kids == 0) {
                        final_score = bk_pos_area_score(&poses[root], prm.komi);
                        state = S_DONE;
                        break;
                    }
                    remaining = prm.rollouts;
                    state = S_SEARCH;
                    break;
                case S_SEARCH: {
                    if (prm.leaves > 1) {                            // the opt-in multi-leaf mode (bk_search_params.leaves)
                        if (search_leaves()) {
                            state = S_WAIT_LEAF;
                            return true;
                        }
                        state = manual ? S_IDLE : S_CHOOSE;
                        break;
                    }
                    bool waiting = false, playing = false;
                    while (remaining > 0) {
                        path.clear();
                        int id = root;
                        path.push_back(id);
                        for (;;) {  // mcts.py:172-183
                            if (nodes[id].n_kids == 0) {
                                if (!nodes[id].expanded && nodes[id].N > prm.expand_thresh) expand(id);
                                break;
                            }
                            id = select(id);
                            path.push_back(id);
                        }
                        if (analyze && path.size() > 2) variations[path[1]].assign(path.begin() + 1, path.end());
                        request_leaf();
                        if (prm.simulate) { playing = true; break; }
                        if (has_request()) { waiting = true; break; }
                        backprop();
                        --remaining;
                    }
                    if (playing) {
                        state = S_PLAYOUT;
                        break;
                    }
                    if (waiting) {
                        add_speculation();
                        state = S_WAIT_LEAF;
                        return true;
                    }
                    state = manual ? S_IDLE : S_CHOOSE;
                    break;
                }
                case S_PLAYOUT:
                    if (pending_expand >= 0) {                       // branch_num: the leaf is expanded before its playout starts
                        if (!nodes[pending_expand].has_prior) {      // (the playout's own nodes must be the newest ones)
                            add_speculation();
                            return true;
                        }
                        const int id = pending_expand;
                        pending_expand = -1;
                        expand(id);
                    }
                    if (!playout()) {
                        add_speculation();
                        return true;
                    }
                    if (has_request()) {                             // the leaf's value (and what travels with it) is still out
                        add_speculation();
                        state = S_WAIT_LEAF;
                        return true;
                    }
                    backprop();
                    --remaining;
                    state = S_SEARCH;
                    break;
                case S_WAIT_LEAF:
                    if (prm.leaves > 1) {
                        back_up_leaves();
                        state = S_SEARCH;
                        break;
                    }
                    if (pending_expand >= 0) {                       // branch_num: the leaf's priors are in: expand it before
                        const int id = pending_expand;               // the next rollout (what it asks for rides with a later request)
                        pending_expand = -1;
                        expand(id);
                    }
                    backprop();
                    --remaining;
                    state = S_SEARCH;
                    break;
                case S_CHOOSE: {
                    const int best = pick_move();
                    note_choice();
                    if (prm.record_visits) {
                        visit_log.emplace_back();
                        const TNode& r = nodes[root];
                        for (int i = 0; i < r.n_kids; ++i) {
                            const TNode& k = nodes[kid_ids[r.kids_off + i]];
                            visit_log.back().emplace_back((int16_t)k.mv, (int32_t)k.N);
                        }
                    }
                    moves.push_back((int16_t)nodes[best].mv);
                    variations.clear();
                    root = best;
                    if (prm.prune) prune();
                    state = S_ROOT_EXPAND;
                    break;
                }
                case S_IDLE:
                    if (remaining > 0) { state = S_SEARCH; break; }
                    // rows asked for by the last rollout's expansion (branch_num: it happens once the priors are in, right
                    // before the search goes idle) still go out: nothing may be left waiting when the caller moves on
                    if (has_request()) return true;
                    return false;
                case S_DONE:
                    return false;
            }
        }
    }

    void deliver_policy(int id, const float* probs) {
        TNode& n = nodes[id];
        if (!n.has_prior) {
            std::vector<double>& dst = transient(id) ? po_priors : priors;
            n.prior_off = (int)dst.size();
            for (int i = 0; i < 81; ++i) dst.push_back((double)probs[i]);
            n.has_prior = 1;
            ++n_policy_evals;
        }
    }
    void deliver_value(int id, float v) {
        TNode& n = nodes[id];
        if (!n.has_value) { n.value = v; n.has_value = 1; ++n_value_evals; }
    }
};

}  // namespace

struct bk_pool {
    double t_advance = 0, t_emit = 0, t_deliver = 0;   // seconds spent in the three phases (bk_pool_phase_seconds)
    std::vector<Game> games;
    int row_cap = 0;              // the smallest `cap` a collect was called with: no single request may outgrow it
    int task_cap = 0, first = 0;  // bk_pool_set_task_cap; the game that goes first into the next batch
    std::vector<int> active;      // games included in the last collect, in batch order
    std::vector<int> pol_off, val_off;
    int threads = 1;
    bool lanes = true;                 // bk_pool_set_lanes: games stay with "their" worker thread (run_lanes)
    std::vector<int> lane_items[64];   // scratch of run_lanes
    // in-batch de-duplication (bk_pool_set_dedup): rows of one batch that are the same position record -- games that are still in
    // the same opening -- travel once; row_of[a] = for the a-th active game the batch row of each of its request rows (policy
    // rows first, then value rows), as laid out by collect_dedup
    bool dedup = false;
    bool mapped = false;          // the batch now out was laid out by collect_dedup (deliver reads row_of)
    std::vector<std::vector<int>> row_of;
    std::vector<bk_pos> uniq_pol, uniq_val;
    uint64_t rows_requested = 0, rows_sent = 0;
};

extern "C" {

void bk_search_params_default(bk_search_params* p) {
    p->rollouts = 400;
    p->expand_thresh = 100;
    p->c_puct = 4.0;
    p->noise_weight = 0.f;
    p->sample_plies = 0;
    p->max_turns = 80;
    p->eager = 1;
    p->eager_top = 0;
    p->komi = 5.5f;
    p->record_visits = 0;
    p->prune = 0;
    p->speculate = 0;
    p->speculate_rows = 128;
    p->request_tasks = 0;
    p->request_steps[0] = p->request_steps[1] = p->request_steps[2] = 0;
    p->branch_num = 0;
    p->simulate = 0;
    p->use_value = 1;
    p->value_weight = 1.0;
    p->leaves = 1;
    p->leaves_visit_only = 0;
}

bk_pool* bk_pool_create(int n_games, const bk_search_params* prm, const uint64_t* seeds, int threads) {
    if (n_games <= 0 || !prm || !seeds) return nullptr;
    bk_pool* p = new bk_pool();
    p->games.reserve(n_games);
    bk_search_params q = *prm;
    if (!q.use_value) q.eager = q.eager_top = 0;                 // no value net (mcts.py:68-69): nothing asks for a value
    if (!q.use_value || q.simulate) q.speculate = 0;             // (a playout's nodes must stay the newest ones of the tree)
    if (q.leaves < 1 || !q.use_value || q.simulate || (q.branch_num > 0 && q.branch_num < 81)) q.leaves = 1;   // (bk_search_params.leaves)
    if (q.leaves > 1) q.speculate = 0;
    q.leaves_visit_only = q.leaves_visit_only != 0;
    for (int i = 0; i < n_games; ++i) p->games.emplace_back(q, seeds[i]);
    p->threads = threads > 0 ? threads : 1;
    return p;
}

void bk_pool_destroy(bk_pool* p) { delete p; }

}  // extern "C"

namespace {

// ---- the host threads of the pools --------------------------------------------------------------------------------------
// A pool step is short: a few hundred games advance for some tens of microseconds each, 500-1000 times per second, between
// two calls from Python.  An OpenMP parallel region per phase spent more time waking its team than working -- on the GPU
// box's EPYC 9575F 16 threads advanced 256 games only 3.3x faster than one, and 32 games SLOWER than one
// (profiles/r03_host_tree.txt).  This team keeps its workers spinning for a while after a job (the next one comes within a
// fraction of a millisecond while a generation runs), lets them sleep when nothing has come for 2 ms, and hands out items
// one by one from an atomic counter (a game that has to expand a node takes 100x longer than one that does not).
// Several callers at once (two pools driven from two Python threads: two engines in one process) each publish their job in a
// slot of their own and the workers serve every slot they find occupied, so the jobs run side by side instead of taking
// turns; the team grows to the helpers all running jobs ask for together.
inline void cpu_relax() {                           // the spin-wait hint of the host's CPU (the library builds on any host)
#if defined(__x86_64__) || defined(__i386__)
    __builtin_ia32_pause();
#elif defined(__aarch64__) || defined(__arm__)
    asm volatile("yield" ::: "memory");
#else
    std::this_thread::yield();
#endif
}
class Team {
public:
    static Team& get() {
        Team* t = instance_.load(std::memory_order_acquire);
        if (t) return *t;
        std::lock_guard<std::mutex> g(create_m());
        t = instance_.load(std::memory_order_acquire);
        if (!t) {
            t = new Team();
            instance_.store(t, std::memory_order_release);
            static bool hooked = false;
            if (!hooked) {
                hooked = true;
                // fork(): the child inherits the team's state but none of its threads (a worker caught between inside++ and
                // inside-- would be waited for for ever, sleepers would be counted that do not exist): the child starts with a
                // fresh team; the old one is left alone (its mutexes may be held by threads that are not there).
                pthread_atfork(nullptr, nullptr, [] { instance_.store(nullptr, std::memory_order_release); new (&create_m()) std::mutex(); });
                std::atexit([] {
                    if (Team* cur = instance_.load(std::memory_order_acquire)) cur->shutdown();
                });
            }
        }
        return *t;
    }
    // 0 on a calling thread, 1.. on the team's workers: stable for the life of the thread
    static int& worker_id() {
        static thread_local int id = 0;
        return id;
    }
    // fn(i) for i in [0, n) on up to `threads` threads (the caller is one of them); returns when all are done.  An exception
    // thrown by fn on any thread is rethrown here, after every item has been handed out.
    void run(int threads, int n, const std::function<void(int)>& fn) {
        if (n <= 0) return;
        if (threads <= 1 || n == 1) {
            for (int i = 0; i < n; ++i) fn(i);
            return;
        }
        const int helpers = std::min(threads - 1, kMaxWorkers);
        Slot* sl = nullptr;
        for (int spins = 0; !sl; ++spins) {                    // a free slot (more than kSlots callers at once: wait for one)
            for (auto& c : slots_) {
                bool free_ = false;
                if (c.taken.compare_exchange_strong(free_, true, std::memory_order_acquire)) { sl = &c; break; }
            }
            if (!sl) { if (spins < 64) cpu_relax(); else std::this_thread::yield(); }
        }
        grow(demand_.fetch_add(helpers, std::memory_order_relaxed) + helpers);
        Job job;
        job.fn = &fn;
        job.n = n;
        job.helpers = helpers;
        job.id = last_id_.fetch_add(1, std::memory_order_relaxed) + 1;
        // Publishing and retiring a job are store-then-load hand-shakes with the workers (Dekker-style): the caller stores
        // cur / announced_ and then reads inside / sleepers_, a worker raises inside / sleepers_ and then reads cur /
        // announced_.  Each side must see the other's store or be seen by it, which only sequentially consistent
        // operations give (with release/acquire the caller's load may pass its own store: a late worker then walked into
        // a job whose stack frame was gone -- a rare segfault in whatever the calling thread did next).
        sl->cur.store(&job, std::memory_order_seq_cst);
        announced_.fetch_add(1, std::memory_order_seq_cst);      // what sleepers watch (they must not look into `job`)
        if (sleepers_.load(std::memory_order_seq_cst) > 0) {
            std::lock_guard<std::mutex> g(m_);
            cv_.notify_all();
        }
        work(job);
        while (job.done.load(std::memory_order_acquire) < n) cpu_relax();
        sl->cur.store(nullptr, std::memory_order_seq_cst);
        while (sl->inside.load(std::memory_order_seq_cst) != 0) cpu_relax();   // nobody still looks at `job`
        demand_.fetch_sub(helpers, std::memory_order_relaxed);
        sl->taken.store(false, std::memory_order_release);
        if (job.failed.load(std::memory_order_acquire)) std::rethrow_exception(job.error);
    }

private:
    static constexpr int kSlots = 8, kMaxWorkers = 63;
    struct Job {
        const std::function<void(int)>* fn = nullptr;
        int n = 0, helpers = 0;
        unsigned long id = 0;
        std::atomic<int> next{0}, done{0}, joined{0};
        std::atomic<bool> failed{false};
        std::exception_ptr error;
    };
    struct alignas(64) Slot {
        std::atomic<Job*> cur{nullptr};
        std::atomic<int> inside{0};
        std::atomic<bool> taken{false};
    };
    static std::mutex& create_m() {
        static std::mutex* m = new std::mutex();    // never destroyed: get() may run during static destruction
        return *m;
    }
    static void work(Job& j) {
        for (;;) {
            const int i = j.next.fetch_add(1, std::memory_order_relaxed);
            if (i >= j.n) break;
            try {
                (*j.fn)(i);
            } catch (...) {                                    // the item counts as done (the caller must not wait for ever);
                bool first = false;                            // the first exception travels to the caller
                if (j.failed.compare_exchange_strong(first, true, std::memory_order_acq_rel)) j.error = std::current_exception();
            }
            j.done.fetch_add(1, std::memory_order_release);
        }
    }
    void grow(int workers) {
        workers = std::min(workers, kMaxWorkers);
        if (n_threads_.load(std::memory_order_acquire) >= workers) return;
        std::lock_guard<std::mutex> g(grow_m_);
        while ((int)th_.size() < workers) {
            const int id = (int)th_.size() + 1;
            th_.emplace_back([this, id] { worker_id() = id; loop(); });
        }
        n_threads_.store((int)th_.size(), std::memory_order_release);
    }
    void shutdown() {
        quit_.store(true);
        {
            std::lock_guard<std::mutex> g(m_);
            cv_.notify_all();
        }
        std::lock_guard<std::mutex> g(grow_m_);
        for (auto& t : th_) t.join();
        th_.clear();
    }
    void loop() {
        unsigned long seen[kSlots] = {0};
        auto idle_since = std::chrono::steady_clock::now();
        int spins = 0;
        bool napping = false;
        while (!quit_.load(std::memory_order_relaxed)) {
            // what the sleep below waits to see CHANGE: taken before the look at the slots, so a job announced after this
            // line keeps the worker awake, and one announced before it is found in its slot (or is retired already).  (The
            // predicate used to be "announced != the last job I joined": a worker that had missed a short job -- asleep, or
            // descheduled under a CPU quota -- then never slept again and spun at 100 % CPU between generations: ADVICE r3.)
            const unsigned long a = announced_.load(std::memory_order_seq_cst);
            bool worked = false;
            for (int k = 0; k < kSlots; ++k) {
                Slot& sl = slots_[k];
                if (!sl.cur.load(std::memory_order_relaxed)) continue;   // a hint only: the hand-shake is below
                sl.inside.fetch_add(1, std::memory_order_seq_cst);
                Job* j = sl.cur.load(std::memory_order_seq_cst);
                if (j && j->id != seen[k]) {
                    seen[k] = j->id;
                    if (j->joined.fetch_add(1, std::memory_order_relaxed) < j->helpers) {
                        work(*j);
                        worked = true;
                    }
                }
                sl.inside.fetch_sub(1, std::memory_order_seq_cst);
            }
            if (worked) {
                idle_since = std::chrono::steady_clock::now();
                spins = 0;
                napping = false;
                continue;
            }
            if (!napping) {                                    // (a nap that merely timed out is followed by the next nap, not by 2 ms of spinning)
                cpu_relax();
                if (++spins < 2000) continue;                  // ~a few microseconds between looks at the clock
                spins = 0;
                if (std::chrono::steady_clock::now() - idle_since < std::chrono::milliseconds(2)) continue;
            }
            {                                                  // nothing for 2 ms: sleep until the next job is announced
                std::unique_lock<std::mutex> g(m_);
                sleepers_.fetch_add(1, std::memory_order_seq_cst);
                // (system_clock: pthread_cond_timedwait, which ThreadSanitizer intercepts -- `make tsan`; a clock step only moves one nap)
                cv_.wait_until(g, std::chrono::system_clock::now() + std::chrono::milliseconds(50),
                               [&] { return quit_.load() || announced_.load(std::memory_order_seq_cst) != a; });
                sleepers_.fetch_sub(1, std::memory_order_seq_cst);
            }
            napping = announced_.load(std::memory_order_seq_cst) == a;   // timed out: look at the slots once and sleep on
            idle_since = std::chrono::steady_clock::now();
        }
    }
    static std::atomic<Team*> instance_;
    Slot slots_[kSlots];
    std::vector<std::thread> th_;
    std::atomic<int> n_threads_{0}, demand_{0};
    std::mutex m_, grow_m_;
    std::condition_variable cv_;
    std::atomic<int> sleepers_{0};
    std::atomic<unsigned long> announced_{0}, last_id_{0};
    std::atomic<bool> quit_{false};
};
std::atomic<Team*> Team::instance_{nullptr};

// A game is worked on by the same thread step after step and phase after phase, so its tree stays in that core's caches: lane
// L = games L, L + T, ... (T = threads taking part); a thread works off its own lane first (its stable id picks it), then helps
// with the others, so a thread that is late or missing costs nothing but the affinity.  Items i in [0, n) belong to game
// game_of(i).  Against handing items out one by one from a single counter: advance phase -10...-18 % on the EPYC host
// (tools/host_tree_bench.py); deliveries, single-threaded before (they touched every tree from the calling thread, i.e. pulled
// it out of its lane's caches), run in the lanes as well: f16x2 self-play 0.50 -> 0.42-0.44 s per 512-game generation.
// bk_pool_set_lanes(p, 0): the single counter (and serial deliveries) -- a switch of the pool, not of the environment: no request
// path of these libraries reads the environment.
template <typename GameOf, typename Fn>
void run_lanes(bk_pool* p, int n, GameOf game_of, Fn fn) {
    const bool lanes = p->lanes;
    const int T = std::min(std::min(p->threads, (int)p->games.size()), 64);
    if (!lanes || T <= 1 || n <= 1) {
        Team::get().run(T, n, fn);
        return;
    }
    for (int l = 0; l < T; ++l) p->lane_items[l].clear();
    for (int i = 0; i < n; ++i) p->lane_items[game_of(i) % T].push_back(i);
    std::atomic<int> cur[64];
    for (int l = 0; l < T; ++l) cur[l].store(0, std::memory_order_relaxed);
    Team::get().run(T, T, [&](int) {
        const int mine = Team::worker_id() % T;
        for (int d = 0; d < T; ++d) {
            const int l = (mine + d) % T;
            const std::vector<int>& items = p->lane_items[l];
            for (;;) {
                const int k = cur[l].fetch_add(1, std::memory_order_relaxed);
                if (k >= (int)items.size()) break;
                fn(items[k]);
            }
        }
    });
}

// advance every game to its next evaluation request and lay the batch out:
// [policy nodes of every game ...][value nodes of every game ...]; games whose request does not fit under
// `cap` keep it for the next collect.  emit(node position, row) writes one row of the batch.
template <typename Emit>
int collect_impl(bk_pool* p, int cap, int* n_policy, Emit emit) {
    const int G = (int)p->games.size();
    std::vector<char> wants(G, 0);
    const auto t0 = std::chrono::steady_clock::now();
    if (p->row_cap == 0 || cap < p->row_cap) {
        // a request grows by speculative rows up to prm.speculate_rows: never beyond what one collect can take (a request
        // that cannot fit would be skipped for ever: ADVICE r2).  A request without speculation is <= 82 rows <= cap.
        p->row_cap = cap;
        for (auto& gm : p->games) {
            gm.prm.speculate_rows = std::min(gm.prm.speculate_rows, cap);
            gm.row_cap = cap;
        }
    }
    auto advance_game = [&](int g) {
        Game& gm = p->games[g];
        if (gm.state != S_DONE && !gm.has_request()) wants[g] = gm.advance() ? 1 : 0;
        else if (gm.has_request()) wants[g] = 1;
    };
    run_lanes(p, G, [](int g) { return g; }, advance_game);
    const auto t1 = std::chrono::steady_clock::now();
    p->t_advance += std::chrono::duration<double>(t1 - t0).count();
    p->active.clear();
    p->pol_off.clear();
    p->val_off.clear();
    p->mapped = false;
    int npol = 0, nval = 0;
    // task_cap: the batch stops growing where the engine's launch would need another round of workgroups (bk_pool_set_task_cap);
    // a game whose request does not fit keeps it and goes first next time
    const int start = p->task_cap > 0 ? p->first % G : 0;
    int first_left = -1;
    for (int k = 0; k < G; ++k) {
        const int g = start + k < G ? start + k : start + k - G;
        if (!wants[g]) continue;
        Game& gm = p->games[g];
        const int need = (int)(gm.req_policy.size() + gm.req_value.size());
        const bool over = p->task_cap > 0 && !p->active.empty() && 2 * npol + nval + gm.request_tasks() > p->task_cap;
        if (npol + nval + need > cap || over) {
            if (first_left < 0) first_left = g;
            continue;
        }
        p->active.push_back(g);
        p->pol_off.push_back(npol);
        p->val_off.push_back(nval);
        npol += (int)gm.req_policy.size();
        nval += (int)gm.req_value.size();
    }
    if (first_left >= 0) p->first = first_left;
    const int A = (int)p->active.size();
    auto emit_game = [&](int a) {
        Game& gm = p->games[p->active[a]];
        for (size_t i = 0; i < gm.req_policy.size(); ++i) emit(&gm.poses[gm.req_policy[i]], (size_t)(p->pol_off[a] + i));
        for (size_t i = 0; i < gm.req_value.size(); ++i) emit(&gm.poses[gm.req_value[i]], (size_t)(npol + p->val_off[a] + i));
        gm.n_requests += 1;
    };
    if (npol + nval >= 64) run_lanes(p, A, [&](int a) { return p->active[a]; }, emit_game);
    else for (int a = 0; a < A; ++a) emit_game(a);
    p->t_emit += std::chrono::duration<double>(std::chrono::steady_clock::now() - t1).count();
    *n_policy = npol;
    return npol + nval;
}

}  // namespace

extern "C" {

int bk_pool_collect(bk_pool* p, uint8_t* feats, int cap, int* n_policy) {
    return collect_impl(p, cap, n_policy, [feats](bk_pos* pos, size_t row) { bk_pos_features_u8(pos, feats + row * 2187, 0); });
}

namespace {
// FNV-1a over the record's 8-byte words (the record is 192 bytes, 8-byte aligned)
inline uint64_t record_hash(const bk_pos& q) {
    const uint64_t* w = reinterpret_cast<const uint64_t*>(&q);
    uint64_t h = 1469598103934665603ull;
    for (size_t i = 0; i < sizeof(bk_pos) / 8; ++i) h = (h ^ w[i]) * 1099511628211ull;
    return h;
}

// bk_pool_collect_pos with in-batch de-duplication.  Two request rows are the same row when their 192-byte records are equal
// byte for byte -- stones, ko, last move, turn AND the (history-dependent) liberty cache: everything the planes are computed
// from -- so the networks' outputs for them are the same bits, whichever game's row is evaluated.  A value row whose record
// also travels as a policy row takes that row's value (a policy row runs both nets).  The batch limits (cap, task_cap) count
// what actually travels.  Per-game counters and records are those of the logical requests: a game cannot tell.
int collect_dedup(bk_pool* p, bk_pos* out, int cap, int* n_policy) {
    const int G = (int)p->games.size();
    std::vector<char> wants(G, 0);
    const auto t0 = std::chrono::steady_clock::now();
    if (p->row_cap == 0 || cap < p->row_cap) {
        p->row_cap = cap;
        for (auto& gm : p->games) {
            gm.prm.speculate_rows = std::min(gm.prm.speculate_rows, cap);
            gm.row_cap = cap;
        }
    }
    run_lanes(p, G, [](int g) { return g; }, [&](int g) {
        Game& gm = p->games[g];
        if (gm.state != S_DONE && !gm.has_request()) wants[g] = gm.advance() ? 1 : 0;
        else if (gm.has_request()) wants[g] = 1;
        if (wants[g]) {                                  // the history-dependent half of features(): see bk_pool_collect_pos
            uint8_t libs[81];
            for (int id : gm.req_policy) bk_pos_liberties(&gm.poses[id], libs);
            for (int id : gm.req_value) bk_pos_liberties(&gm.poses[id], libs);
        }
    });
    const auto t1 = std::chrono::steady_clock::now();
    p->t_advance += std::chrono::duration<double>(t1 - t0).count();
    p->active.clear();
    p->row_of.clear();
    p->uniq_pol.clear();
    p->uniq_val.clear();
    // record hash -> index into uniq_pol (kind 0) / uniq_val (kind 1); collisions resolved by comparing the records
    std::unordered_multimap<uint64_t, std::pair<int, int>> index;
    auto find = [&](const bk_pos& q, uint64_t h, int kind) {
        auto range = index.equal_range(h);
        for (auto it = range.first; it != range.second; ++it)
            if (it->second.first == kind &&
                std::memcmp(&(kind ? p->uniq_val : p->uniq_pol)[it->second.second], &q, sizeof(bk_pos)) == 0)
                return it->second.second;
        return -1;
    };
    const int start = p->task_cap > 0 ? p->first % G : 0;
    int first_left = -1;
    std::vector<int> rows;      // of the game being placed: >= 0 policy index, < 0: -(value index) - 1
    std::vector<std::pair<uint64_t, std::pair<int, int>>> added;
    for (int k = 0; k < G; ++k) {
        const int g = start + k < G ? start + k : start + k - G;
        if (!wants[g]) continue;
        Game& gm = p->games[g];
        rows.clear();
        added.clear();
        const size_t np0 = p->uniq_pol.size(), nv0 = p->uniq_val.size();
        for (int id : gm.req_policy) {
            const bk_pos& q = gm.poses[id];
            const uint64_t h = record_hash(q);
            int at = find(q, h, 0);
            if (at < 0) {
                at = (int)p->uniq_pol.size();
                p->uniq_pol.push_back(q);
                index.emplace(h, std::make_pair(0, at));
                added.push_back({h, {0, at}});
            }
            rows.push_back(at);
        }
        for (int id : gm.req_value) {
            const bk_pos& q = gm.poses[id];
            const uint64_t h = record_hash(q);
            int at = find(q, h, 0);
            if (at >= 0) { rows.push_back(at); continue; }       // travels as a policy row: its value comes with it
            at = find(q, h, 1);
            if (at < 0) {
                at = (int)p->uniq_val.size();
                p->uniq_val.push_back(q);
                index.emplace(h, std::make_pair(1, at));
                added.push_back({h, {1, at}});
            }
            rows.push_back(-at - 1);
        }
        const int npol = (int)p->uniq_pol.size(), nval = (int)p->uniq_val.size();
        const bool over = p->task_cap > 0 && !p->active.empty() && 2 * npol + nval > p->task_cap;
        if (npol + nval > cap || over) {                     // does not fit: take its new rows out again
            for (auto& a : added) {
                auto range = index.equal_range(a.first);
                for (auto it = range.first; it != range.second; ++it)
                    if (it->second == a.second) { index.erase(it); break; }
            }
            p->uniq_pol.resize(np0);
            p->uniq_val.resize(nv0);
            if (first_left < 0) first_left = g;
            continue;
        }
        p->active.push_back(g);
        p->row_of.push_back(rows);
        p->rows_requested += rows.size();
        gm.n_requests += 1;
    }
    if (first_left >= 0) p->first = first_left;
    const int npol = (int)p->uniq_pol.size(), nval = (int)p->uniq_val.size();
    if (npol) std::memcpy(out, p->uniq_pol.data(), (size_t)npol * sizeof(bk_pos));
    if (nval) std::memcpy(out + npol, p->uniq_val.data(), (size_t)nval * sizeof(bk_pos));
    for (auto& rows_a : p->row_of)
        for (int& r : rows_a)
            if (r < 0) r = npol + (-r - 1);
    p->rows_sent += (uint64_t)(npol + nval);
    p->mapped = true;
    p->t_emit += std::chrono::duration<double>(std::chrono::steady_clock::now() - t1).count();
    *n_policy = npol;
    return npol + nval;
}
}  // namespace

void bk_pool_set_dedup(bk_pool* p, int on) { p->dedup = on != 0; }
void bk_pool_set_lanes(bk_pool* p, int on) { p->lanes = on != 0; }
void bk_pool_dedup_rows(const bk_pool* p, uint64_t* requested, uint64_t* sent) {
    *requested = p->rows_requested;
    *sent = p->rows_sent;
}

int bk_pool_collect_pos(bk_pool* p, bk_pos* out, int cap, int* n_policy) {
    if (p->dedup) return collect_dedup(p, out, cap, n_policy);
    // the history-dependent half of nnet.features() -- the lazy liberty-cache refresh, go.py:220-243 -- runs
    // here, on the node itself (its children inherit the refreshed cache, as with bk_pool_collect); the
    // planes are then a pure function of the record and are computed by the consumer (the GPU encoder)
    return collect_impl(p, cap, n_policy, [out](bk_pos* pos, size_t row) {
        uint8_t libs[81];
        bk_pos_liberties(pos, libs);
        out[row] = *pos;
    });
}

// Stress of the worker team (tests/test_selfplay_cpu.py): `jobs` parallel regions of a few items each, every so often after a
// pause long enough for the workers to fall asleep, so that workers keep arriving late at regions that are being retired.
// A region's bookkeeping lives on the caller's stack; after every region the same stack area is filled with a pattern and
// checked a little later -- a worker that still walked into the retired region would change it.
// Returns 0 when every item of every region ran exactly once and no pattern was touched.
namespace {
__attribute__((noinline)) int team_region(int threads, int n) {
    std::atomic<int> hits[8];
    for (auto& h : hits) h.store(0, std::memory_order_relaxed);
    Team::get().run(threads, n, [&](int i) { hits[i].fetch_add(1, std::memory_order_relaxed); });
    for (int i = 0; i < n; ++i)
        if (hits[i].load(std::memory_order_relaxed) != 1) return 1;
    return 0;
}
__attribute__((noinline)) int team_canary(int spins) {
    volatile unsigned char pad[768];
    for (auto& b : pad) b = 0xA5;
    for (int i = 0; i < spins; ++i) cpu_relax();
    for (auto& b : pad)
        if (b != 0xA5) return 1;
    return 0;
}
}  // namespace
int bk_team_selftest(int threads, int jobs) {
    for (int k = 0; k < jobs; ++k) {
        if (team_region(threads, 1 + k % 7)) return 1 + k;
        if (team_canary(k % 64)) return -(1 + k);
        if (k % 4096 == 4095) std::this_thread::sleep_for(std::chrono::milliseconds(3));
    }
    return 0;
}

// Two callers at once: each runs one region whose items wait (bounded) until they have seen the OTHER caller's region running.
// A team that lets its callers take turns cannot finish this: returns 0 when both regions met, 1 when they did not within
// `timeout_ms` (the regions then give up, so the call always returns).
int bk_team_selftest_concurrent(int threads, int timeout_ms) {
    std::atomic<int> up[2] = {{0}, {0}}, met[2] = {{0}, {0}};
    auto caller = [&](int me) {
        Team::get().run(threads, threads, [&](int) {
            up[me].store(1, std::memory_order_seq_cst);
            const auto t0 = std::chrono::steady_clock::now();
            while (!up[1 - me].load(std::memory_order_seq_cst)) {
                if (std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(timeout_ms)) return;
                cpu_relax();
            }
            met[me].store(1, std::memory_order_seq_cst);
        });
    };
    std::thread a(caller, 0), b(caller, 1);
    a.join();
    b.join();
    return met[0].load() && met[1].load() ? 0 : 1;
}

void bk_pool_phase_seconds(const bk_pool* p, double* out3) {   // advance, emit, deliver
    out3[0] = p->t_advance;
    out3[1] = p->t_emit;
    out3[2] = p->t_deliver;
}

void bk_pool_deliver(bk_pool* p, const float* probs, const float* values) {
    const auto t0 = std::chrono::steady_clock::now();
    int npol = 0, nval = 0;
    for (size_t a = 0; a < p->active.size(); ++a) {
        npol += (int)p->games[p->active[a]].req_policy.size();
        nval += (int)p->games[p->active[a]].req_value.size();
    }
    const int A = (int)p->active.size();
    const bool mapped = p->mapped;                                          // the batch was laid out by collect_dedup
    auto deliver_game = [&](int a) {
        Game& gm = p->games[p->active[a]];
        if (mapped) {
            const std::vector<int>& rows = p->row_of[a];
            const size_t np_ = gm.req_policy.size();
            for (size_t i = 0; i < np_; ++i) {
                gm.deliver_policy(gm.req_policy[i], probs + (size_t)rows[i] * 81);
                if (values) gm.deliver_value(gm.req_policy[i], values[rows[i]]);
            }
            for (size_t i = 0; i < gm.req_value.size(); ++i) gm.deliver_value(gm.req_value[i], values[rows[np_ + i]]);
            gm.req_policy.clear();
            gm.req_value.clear();
            return;
        }
        for (size_t i = 0; i < gm.req_policy.size(); ++i) {
            const int row = p->pol_off[a] + (int)i;
            gm.deliver_policy(gm.req_policy[i], probs + (size_t)row * 81);
            if (values) gm.deliver_value(gm.req_policy[i], values[row]);
        }
        for (size_t i = 0; i < gm.req_value.size(); ++i) gm.deliver_value(gm.req_value[i], values[npol + p->val_off[a] + (int)i]);
        gm.req_policy.clear();
        gm.req_value.clear();
    };
    // in the games' lanes (with a single hand-out counter a parallel delivery cost more than it saved; bk_pool_set_lanes(p, 0) keeps it serial)
    const bool lanes = p->lanes;
    if (lanes && npol + nval >= 64) run_lanes(p, A, [&](int a) { return p->active[a]; }, deliver_game);
    else for (int a = 0; a < A; ++a) deliver_game(a);
    p->active.clear();
    p->row_of.clear();
    p->t_deliver += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
}

int bk_pool_n_games(const bk_pool* p) { return (int)p->games.size(); }

int bk_pool_n_done(const bk_pool* p) {
    int n = 0;
    for (const auto& g : p->games) n += g.state == S_DONE;
    return n;
}

int bk_pool_game_info(const bk_pool* p, int g, bk_game_info* out) {
    if (g < 0 || g >= (int)p->games.size() || !out) return -1;
    const Game& gm = p->games[g];
    out->done = gm.state == S_DONE;
    out->n_moves = (int)gm.moves.size();
    out->score = gm.final_score;
    out->n_value_evals = gm.n_value_evals;
    out->n_policy_evals = gm.n_policy_evals;
    out->n_requests = gm.n_requests;
    out->n_nodes = (int)gm.nodes.size();
    out->root_N = gm.root >= 0 ? gm.nodes[gm.root].N : 0;
    out->root_V = gm.root >= 0 ? gm.nodes[gm.root].V : 0.0;
    return 0;
}

int bk_pool_game_moves(const bk_pool* p, int g, int16_t* out, int cap) {
    if (g < 0 || g >= (int)p->games.size()) return -1;
    const auto& m = p->games[g].moves;
    const int n = std::min((int)m.size(), cap);
    if (n > 0) std::memcpy(out, m.data(), (size_t)n * sizeof(int16_t));   // (an empty vector's data() may be null)
    return (int)m.size();
}

int bk_pool_game_visits(const bk_pool* p, int g, int ply, int16_t* moves, int32_t* N) {
    if (g < 0 || g >= (int)p->games.size()) return -1;
    const auto& log = p->games[g].visit_log;
    if (ply < 0 || ply >= (int)log.size()) return -1;
    for (size_t i = 0; i < log[ply].size(); ++i) {
        moves[i] = log[ply][i].first;
        N[i] = log[ply][i].second;
    }
    return (int)log[ply].size();
}

/* ---- manual control: the single-tree MCTS surface (rollout / choose / set_root) on a pool game ---- */
void bk_pool_set_manual(bk_pool* p, int on) {
    for (auto& g : p->games) g.manual = on != 0;
}

void bk_pool_set_task_cap(bk_pool* p, int tasks) { p->task_cap = tasks > 0 ? tasks : 0; }

void bk_pool_set_speculation(bk_pool* p, int speculate, int rows, int request_tasks) {   // see bk_search_params
    for (auto& g : p->games) {
        g.prm.speculate = speculate;
        g.prm.speculate_rows = p->row_cap > 0 ? std::min(rows, p->row_cap) : rows;
        g.prm.request_tasks = request_tasks;
        g.prm.request_steps[0] = request_tasks;   // the later steps stay as created
    }
}

int bk_pool_add_rollouts(bk_pool* p, int g, int n) {
    if (g < 0 || g >= (int)p->games.size() || n < 0) return -1;
    p->games[g].remaining += n;
    return p->games[g].remaining;
}

int bk_pool_choose(bk_pool* p, int g) {  // MCTS.choose at the root (mcts.py:110-131); returns the move, BK_NO_MOVE if none
    if (g < 0 || g >= (int)p->games.size()) return BK_NO_MOVE;
    Game& gm = p->games[g];
    if (gm.root < 0 || gm.has_request() || gm.nodes[gm.root].terminal || gm.nodes[gm.root].n_kids == 0) return BK_NO_MOVE;
    const int best = gm.pick_move();
    gm.note_choice();
    gm.moves.push_back((int16_t)gm.nodes[best].mv);
    const int mv = gm.nodes[best].mv;
    gm.reroot(best);
    return mv;
}

int bk_pool_play(bk_pool* p, int g, int move) {  // make `move` (or BK_PASS) the new root; 0 or a BK_ILLEGAL_* code
    if (g < 0 || g >= (int)p->games.size()) return -1;
    Game& gm = p->games[g];
    if (gm.has_request()) return -1;
    if (gm.root < 0) { bk_pos q; bk_pos_init(&q); gm.root = gm.intern(q); }
    bk_pos q = gm.poses[gm.root];
    const int rc = bk_pos_play(&q, move);
    if (rc) return rc;
    gm.moves.push_back((int16_t)move);
    gm.reroot(gm.intern(q));
    return 0;
}

int bk_pool_set_position(bk_pool* p, int g, const bk_pos* pos) {  // new root from an arbitrary position (clear_board, handicap)
    if (g < 0 || g >= (int)p->games.size() || !pos) return -1;
    Game& gm = p->games[g];
    if (gm.has_request()) return -1;
    gm.moves.clear();
    gm.reroot(gm.intern(*pos));
    return 0;
}

int bk_pool_root_pos(const bk_pool* p, int g, bk_pos* out) {
    if (g < 0 || g >= (int)p->games.size() || p->games[g].root < 0) return -1;
    *out = p->games[g].poses[p->games[g].root];
    return 0;
}

/* ---- read-only views of the tree (the reference keeps N / V / children as dicts anybody may read, mcts.py:46-52) ---- */
void bk_pool_set_analyze(bk_pool* p, int on) {
    for (auto& g : p->games) {
        g.analyze = on != 0;
        g.variations.clear();
    }
}

int bk_pool_variation(const bk_pool* p, int g, int move, int32_t* ids, int cap) {
    if (g < 0 || g >= (int)p->games.size()) return -1;
    const Game& gm = p->games[g];
    if (gm.root < 0) return 0;
    const TNode& r = gm.nodes[gm.root];
    for (int i = 0; i < r.n_kids; ++i) {
        const int c = gm.kid_ids[r.kids_off + i];
        if (gm.nodes[c].mv != move) continue;
        auto it = gm.variations.find(c);
        if (it == gm.variations.end()) return 0;
        const int n = std::min((int)it->second.size(), cap);
        for (int k = 0; k < n; ++k) ids[k] = it->second[k];
        return (int)it->second.size();
    }
    return 0;
}

int bk_pool_find(const bk_pool* p, int g, const bk_pos* pos) {
    if (g < 0 || g >= (int)p->games.size() || !pos) return -1;
    return p->games[g].find(*pos);
}

int bk_pool_root_id(const bk_pool* p, int g) { return (g < 0 || g >= (int)p->games.size()) ? -1 : p->games[g].root; }

int bk_pool_node(const bk_pool* p, int g, int id, bk_node_info* out, bk_pos* pos) {
    if (g < 0 || g >= (int)p->games.size()) return -1;
    const Game& gm = p->games[g];
    if (id < 0 || id >= (int)gm.nodes.size()) return -1;
    const TNode& n = gm.nodes[id];
    if (out) {
        out->N = n.N;
        out->n_children = n.n_kids;
        out->V = n.V;
        out->value = n.value;
        out->move = (int16_t)n.mv;
        out->flags = (uint16_t)((n.expanded ? BK_NODE_EXPANDED : 0) | (n.terminal ? BK_NODE_TERMINAL : 0) |
                                (n.has_value ? BK_NODE_HAS_VALUE : 0) | (n.has_prior ? BK_NODE_HAS_PRIOR : 0));
    }
    if (pos) *pos = gm.poses[id];
    return 0;
}

int bk_pool_node_q(const bk_pool* p, int g, int id, double* q) {
    if (g < 0 || g >= (int)p->games.size() || !q) return -1;
    const Game& gm = p->games[g];
    if (id < 0 || id >= (int)gm.nodes.size()) return -1;
    *q = gm.prm.simulate ? gm.Qs[id] : 0.0;
    return 0;
}

int bk_pool_node_children(const bk_pool* p, int g, int id, int32_t* ids, int cap) {
    if (g < 0 || g >= (int)p->games.size()) return -1;
    const Game& gm = p->games[g];
    if (id < 0 || id >= (int)gm.nodes.size()) return -1;
    const TNode& n = gm.nodes[id];
    for (int i = 0; i < n.n_kids && i < cap; ++i) ids[i] = gm.kid_ids[n.kids_off + i];
    return n.n_kids;
}

int bk_pool_node_prior(const bk_pool* p, int g, int id, double* prior) {
    if (g < 0 || g >= (int)p->games.size() || !prior) return -1;
    const Game& gm = p->games[g];
    if (id < 0 || id >= (int)gm.nodes.size() || !gm.nodes[id].has_prior) return -1;
    std::memcpy(prior, &gm.priors[gm.nodes[id].prior_off], 81 * sizeof(double));
    return 0;
}

int bk_pool_principal_variation(const bk_pool* p, int g, int16_t* moves, int cap) {
    if (g < 0 || g >= (int)p->games.size()) return -1;
    const Game& gm = p->games[g];
    int id = gm.root, n = 0;
    while (id >= 0 && gm.nodes[id].n_kids > 0 && n < cap) {   // most visited child at every level, lowest move on ties
        const TNode& nd = gm.nodes[id];
        int best = -1, best_n = 0;
        for (int i = 0; i < nd.n_kids; ++i) {
            const int c = gm.kid_ids[nd.kids_off + i];
            if (gm.nodes[c].N > best_n) { best = c; best_n = gm.nodes[c].N; }
        }
        if (best < 0) break;
        moves[n++] = (int16_t)gm.nodes[best].mv;
        id = best;
    }
    return n;
}

int bk_pool_root_children(const bk_pool* p, int g, int16_t* moves, int32_t* N, double* V) {
    if (g < 0 || g >= (int)p->games.size()) return -1;
    const Game& gm = p->games[g];
    if (gm.root < 0) return 0;
    const TNode& r = gm.nodes[gm.root];
    for (int i = 0; i < r.n_kids; ++i) {
        const TNode& k = gm.nodes[gm.kid_ids[r.kids_off + i]];
        moves[i] = (int16_t)k.mv;
        N[i] = k.N;
        V[i] = k.V;
    }
    return r.n_kids;
}


int bk_pool_game_stats(const bk_pool* p, int g, bk_game_stats* out) {
    if (g < 0 || g >= (int)p->games.size() || !out) return -1;
    const Game& gm = p->games[g];
    std::memcpy(out->root_visits, gm.root_visits, sizeof gm.root_visits);
    out->sum_root_value = (double)gm.sum_root_value_q / 4294967296.0;          // exact: |q| < 2^53
    out->sum_abs_root_value = (double)gm.sum_abs_root_value_q / 4294967296.0;
    out->n_root_values = gm.n_root_values;
    return 0;
}

/* ---- snapshot / restore of one game's tree (MCTS.__getstate__ / __setstate__ / __deepcopy__, mcts.py:81-108) ---- */
}  // extern "C"

namespace {
constexpr uint32_t kSnapMagic = 0x31544B42u;   // "BKT1"
constexpr uint32_t kSnapVersion = 2;   // 2: bk_search_params.leaves, the multi-leaf mode's waiting rollouts
struct SnapWriter {
    std::vector<uint8_t> b;
    void raw(const void* p, size_t n) { const uint8_t* q = static_cast<const uint8_t*>(p); b.insert(b.end(), q, q + n); }
    template <typename T> void pod(const T& v) { raw(&v, sizeof(T)); }
    template <typename T> void vec(const std::vector<T>& v) {
        const uint64_t n = v.size();
        pod(n);
        if (n) raw(v.data(), n * sizeof(T));
    }
    void map(const std::unordered_map<int, std::vector<int>>& m) {
        std::vector<int> keys;
        for (const auto& kv : m) keys.push_back(kv.first);
        std::sort(keys.begin(), keys.end());                 // equal trees give equal bytes
        const uint64_t n = keys.size();
        pod(n);
        for (int k : keys) { pod(k); vec(m.at(k)); }
    }
};
struct SnapReader {
    const uint8_t* p;
    const uint8_t* end;
    bool ok = true;
    bool raw(void* dst, size_t n) {
        if (!ok || (size_t)(end - p) < n) return ok = false;
        std::memcpy(dst, p, n);
        p += n;
        return true;
    }
    template <typename T> bool pod(T& v) { return raw(&v, sizeof(T)); }
    template <typename T> bool vec(std::vector<T>& v) {
        uint64_t n = 0;
        if (!pod(n) || n > (uint64_t)(end - p) / sizeof(T)) return ok = false;
        v.resize((size_t)n);
        return n == 0 || raw(v.data(), (size_t)n * sizeof(T));
    }
    bool map(std::unordered_map<int, std::vector<int>>& m) {
        uint64_t n = 0;
        if (!pod(n) || n > (uint64_t)(end - p)) return ok = false;
        m.clear();
        for (uint64_t i = 0; i < n && ok; ++i) {
            int k = 0;
            std::vector<int> v;
            if (pod(k) && vec(v)) m.emplace(k, std::move(v));
        }
        return ok;
    }
};

void snapshot_game(const Game& gm, SnapWriter& w) {
    w.pod(kSnapMagic);
    w.pod(kSnapVersion);
    const uint32_t sizes[3] = {(uint32_t)sizeof(TNode), (uint32_t)sizeof(bk_pos), (uint32_t)sizeof(bk_search_params)};
    w.pod(sizes);
    w.pod(gm.prm);
    w.vec(gm.nodes);
    w.vec(gm.poses);
    w.vec(gm.Qs);
    w.vec(gm.kid_ids);
    w.vec(gm.priors);
    const int32_t ints[8] = {gm.root, (int32_t)gm.state, gm.remaining, gm.pending_expand, gm.po, gm.po_mark, gm.po_reward, gm.row_cap};
    w.pod(ints);
    // (the last rollout's path is read again only while its leaf's value is out or its playout runs; after a re-rooting of a
    // pruning tree its ids are stale -- every rollout starts with path.clear())
    w.vec(gm.prm.leaves <= 1 && (gm.state == S_WAIT_LEAF || gm.state == S_PLAYOUT) ? gm.path : std::vector<int>());
    w.vec(gm.pend);                                      // (the multi-leaf mode's waiting rollouts: [length, ids ...] each)
    w.pod((int32_t)gm.n_pend);
    w.vec(gm.req_policy);
    w.vec(gm.req_value);
    w.vec(gm.spec_queue);
    w.vec(gm.spill);
    w.map(gm.spec_kids);
    w.map(gm.variations);
    w.vec(gm.po_priors);
    w.pod(gm.rng.s);
    w.vec(gm.moves);
    const uint64_t nlog = gm.visit_log.size();
    w.pod(nlog);
    for (const auto& ply : gm.visit_log) w.vec(ply);
    const uint64_t cnt[3] = {gm.n_value_evals, gm.n_policy_evals, gm.n_requests};
    w.pod(cnt);
    w.pod(gm.final_score);
    const uint8_t flags[2] = {(uint8_t)gm.manual, (uint8_t)gm.analyze};
    w.pod(flags);
    w.pod(gm.root_visits);
    const int64_t vq[2] = {gm.sum_root_value_q, gm.sum_abs_root_value_q};
    w.pod(vq);
    w.pod(gm.n_root_values);
}

// false: the bytes are not a snapshot this build can take (the game is left untouched)
bool restore_game(Game& dst, const uint8_t* buf, size_t len) {
    SnapReader r{buf, buf + len};
    uint32_t magic = 0, version = 0, sizes[3] = {0, 0, 0};
    if (!r.pod(magic) || !r.pod(version) || !r.pod(sizes) || magic != kSnapMagic || version != kSnapVersion ||
        sizes[0] != sizeof(TNode) || sizes[1] != sizeof(bk_pos) || sizes[2] != sizeof(bk_search_params))
        return false;
    bk_search_params prm;
    if (!r.pod(prm)) return false;
    Game gm(prm, 0);
    int32_t ints[8];
    r.vec(gm.nodes); r.vec(gm.poses); r.vec(gm.Qs); r.vec(gm.kid_ids); r.vec(gm.priors);
    if (!r.pod(ints)) return false;
    if (ints[1] < (int32_t)S_INIT || ints[1] > (int32_t)S_IDLE) return false;          // (checked as an integer: an enum must not hold anything else)
    gm.root = ints[0]; gm.state = (State)ints[1]; gm.remaining = ints[2]; gm.pending_expand = ints[3];
    gm.po = ints[4]; gm.po_mark = ints[5]; gm.po_reward = ints[6]; gm.row_cap = ints[7];
    int32_t n_pend = 0;
    r.vec(gm.path); r.vec(gm.pend); r.pod(n_pend); r.vec(gm.req_policy); r.vec(gm.req_value); r.vec(gm.spec_queue); r.vec(gm.spill);
    r.map(gm.spec_kids); r.map(gm.variations); r.vec(gm.po_priors);
    r.pod(gm.rng.s);
    r.vec(gm.moves);
    uint64_t nlog = 0;
    if (!r.pod(nlog) || nlog > len) return false;
    gm.visit_log.resize((size_t)nlog);
    for (auto& ply : gm.visit_log) r.vec(ply);
    uint64_t cnt[3];
    uint8_t flags[2];
    int64_t vq[2];
    r.pod(cnt); r.pod(gm.final_score); r.pod(flags); r.pod(gm.root_visits); r.pod(vq); r.pod(gm.n_root_values);
    if (!r.ok || r.p != r.end) return false;
    gm.n_value_evals = cnt[0]; gm.n_policy_evals = cnt[1]; gm.n_requests = cnt[2];
    gm.manual = flags[0] != 0; gm.analyze = flags[1] != 0;
    gm.sum_root_value_q = vq[0]; gm.sum_abs_root_value_q = vq[1];
    // every index the search will follow must point inside the arrays it indexes
    const int n = (int)gm.nodes.size();
    auto node_ok = [&](int id) { return id >= 0 && id < n; };
    auto all_ok = [&](const std::vector<int>& v) { return std::all_of(v.begin(), v.end(), node_ok); };
    if (gm.poses.size() != gm.nodes.size() || (gm.prm.simulate ? gm.Qs.size() != gm.nodes.size() : !gm.Qs.empty())) return false;
    if (gm.state < S_INIT || gm.state > S_IDLE || gm.remaining < 0) return false;
    // the search parameters and the request limit travel with the game: what the step loop divides by, indexes with or sizes
    // requests from must be in range (a patched row_cap of -5 or a wild branch_num was taken as it came)
    {
        const bk_search_params& q = gm.prm;
        auto in = [](long v, long lo, long hi) { return v >= lo && v <= hi; };
        auto fin = [](double v, double lo, double hi) { return v >= lo && v <= hi; };          // (false for a NaN)
        if (!in(q.rollouts, 0, 1 << 24) || !in(q.expand_thresh, 0, 1 << 30) || !fin(q.c_puct, 0.0, 1e6) || !fin(q.noise_weight, 0.0, 1.0) ||
            !in(q.sample_plies, 0, 1 << 20) || !in(q.max_turns, 0, 100000) || !in(q.eager, 0, 1) || !fin(q.komi, -1000.0, 1000.0) ||
            !in(q.record_visits, 0, 1) || !in(q.prune, 0, 1) || !in(q.speculate, 0, 1 << 30) || !in(q.speculate_rows, 0, 1 << 30) ||
            !in(q.request_tasks, 0, 1 << 30) || !in(q.eager_top, 0, 81) || !in(q.branch_num, 0, 81) || !in(q.simulate, 0, 1) ||
            !in(q.use_value, 0, 1) || !fin(q.value_weight, 0.0, 1.0) || !in(q.leaves, 1, 64) || !in(q.leaves_visit_only, 0, 1))
            return false;
        if (q.leaves > 1 && (q.simulate || !q.use_value || (q.branch_num > 0 && q.branch_num < 81) || q.speculate)) return false;   // (bk_pool_create's rule)
        for (int v : q.request_steps)
            if (!in(v, 0, 1 << 30)) return false;
        if (!q.use_value && !q.simulate) return false;                       // (bk_search_params: no value net needs simulate)
        if (gm.row_cap < 1) return false;
    }
    if (n == 0 ? gm.root != -1 : !node_ok(gm.root)) return false;
    if (gm.pending_expand != -1 && !node_ok(gm.pending_expand)) return false;
    if (gm.po != -1 || gm.po_mark != -1 || !gm.po_priors.empty()) return false;       // never in the middle of a playout
    if (!gm.req_policy.empty() || !gm.req_value.empty()) return false;                // ... nor with a request out (bk_pool_snapshot refuses both)
    if (!all_ok(gm.kid_ids) || !all_ok(gm.path) || !all_ok(gm.req_policy) || !all_ok(gm.req_value) || !all_ok(gm.spec_queue) || !all_ok(gm.spill)) return false;
    for (const auto* m : {&gm.spec_kids, &gm.variations})
        for (const auto& kv : *m)
            if (!node_ok(kv.first) || !all_ok(kv.second)) return false;
    if (gm.priors.size() % 81) return false;
    for (int i = 0; i < n; ++i) {                        // the rules code indexes tables by stone colour, point and ko point
        const bk_pos& q = gm.poses[(size_t)i];
        for (int k = 0; k < 81; ++k)
            if (q.board[k] != BK_EMPTY && q.board[k] != BK_BLACK && q.board[k] != BK_WHITE) return false;
        if (q.ko < BK_NO_KO || q.ko > 80 || q.turn < 0 || q.turn > 100000) return false;
        if (!(q.last_move == BK_NO_MOVE || q.last_move == BK_PASS || (q.last_move >= 0 && q.last_move <= 80))) return false;
        if (gm.nodes[(size_t)i].mv != q.last_move) return false;      // intern(): a node is reached by its position's last move
    }
    // the statistics the search and the end-of-generation sums compute with: visit counts that can be summed in an int, value sums
    // no larger than the visits behind them (|value| <= 1 per backup; + the virtual losses of the multi-leaf mode), finite averages --
    // a wild V would overflow the fixed-point root-value sums (UBSan on an accepted corruption, round 6)
    for (const TNode& nd : gm.nodes) {
        if (nd.N < 0 || nd.N > (1 << 28) || !(std::fabs(nd.V) <= (double)nd.N + 64.0) || !(std::fabs(nd.avg) <= 65.0) || !(std::fabs(nd.value) <= 1.0f)) return false;
    }
    if (gm.n_root_values > ((uint64_t)1 << 30) || std::llabs(gm.sum_root_value_q) > (long long)((gm.n_root_values + 1) << 32) ||
        gm.sum_abs_root_value_q < 0 || gm.sum_abs_root_value_q > (long long)((gm.n_root_values + 1) << 32)) return false;
    for (size_t i = 0; i < gm.Qs.size(); ++i)
        if (!(std::fabs(gm.Qs[i]) <= (double)gm.nodes[i].N + 1.0)) return false;
    for (double pr : gm.priors)
        if (!(pr >= 0.0 && pr <= 1.0e6)) return false;
    for (const TNode& nd : gm.nodes) {
        if (nd.n_kids < 0 || nd.kids_off < 0 || (size_t)nd.kids_off + (size_t)nd.n_kids > gm.kid_ids.size()) return false;
        if (nd.has_prior && (nd.prior_off < 0 || (size_t)nd.prior_off + 81 > gm.priors.size())) return false;
        // a node with children is selected through its priors and its children's moves: both must be there (no request is out
        // when a snapshot is taken, so no expansion is waiting for its policy row)
        if (nd.n_kids > 0 && !nd.has_prior) return false;
        for (int k = 0; k < nd.n_kids; ++k) {
            const int mv = gm.nodes[gm.kid_ids[nd.kids_off + k]].mv;
            if (mv < 0 || mv >= 81) return false;
        }
    }
    // state and path belong together: the states that come back to the last rollout (its leaf's value is out, its playout runs) walk
    // `path` from the root down -- it must be there, start at the root and follow the tree's own edges; every other state starts
    // its next rollout with path.clear() and the writer leaves it out (a snapshot patched to S_WAIT_LEAF with an empty
    // path was accepted, and the next collect read path.back())
    auto walks_the_tree = [&](const int* pth, size_t len) {
        if (len == 0 || pth[0] != gm.root) return false;
        for (size_t i = 0; i + 1 < len; ++i) {
            const TNode& nd = gm.nodes[(size_t)pth[i]];
            const int* k0 = gm.kid_ids.data() + nd.kids_off;
            if (std::find(k0, k0 + nd.n_kids, pth[i + 1]) == k0 + nd.n_kids) return false;
        }
        return true;
    };
    if (gm.prm.leaves > 1) {
        // the multi-leaf mode keeps its waiting rollouts in `pend` ([length, ids ...] each, n_pend of them, at most `leaves`) and uses
        // `path` as scratch: a snapshot has them in S_WAIT_LEAF only, each walking the tree from the root, and their virtual losses on
        // the nodes (a node's visit count covers the losses it carries: taking them off must not go below zero)
        if (!gm.path.empty() || n_pend < 0 || n_pend > gm.prm.leaves || (gm.state != S_WAIT_LEAF && (n_pend != 0 || !gm.pend.empty()))) return false;
        std::vector<int> losses(gm.nodes.size(), 0);
        size_t at = 0;
        for (int r = 0; r < n_pend; ++r) {
            if (at >= gm.pend.size()) return false;
            const int len = gm.pend[at++];
            if (len < 1 || (size_t)len > gm.pend.size() - at) return false;
            for (int i = 0; i < len; ++i)
                if (!node_ok(gm.pend[at + (size_t)i])) return false;
            if (!walks_the_tree(gm.pend.data() + at, (size_t)len)) return false;
            for (int i = 1; i < len; ++i) losses[(size_t)gm.pend[at + (size_t)i]] += 1;
            if (!gm.nodes[(size_t)gm.pend[at + (size_t)len - 1]].has_value) return false;     // backprop reads the leaf's value
            at += (size_t)len;
        }
        if (at != gm.pend.size()) return false;
        for (size_t i = 0; i < gm.nodes.size(); ++i)
            if (gm.nodes[i].N < losses[i]) return false;
        gm.n_pend = n_pend;
    } else if (gm.state == S_WAIT_LEAF || gm.state == S_PLAYOUT) {
        if (!gm.pend.empty() || n_pend != 0) return false;
        if (!walks_the_tree(gm.path.data(), gm.path.size())) return false;
        if (gm.state == S_PLAYOUT && !gm.prm.simulate) return false;
    } else if (!gm.path.empty() || !gm.pend.empty() || n_pend != 0) {
        return false;
    }
    // states that index nodes[root] need a root (and what they read of it); a pending expansion is looked at in the waiting states only
    if (n == 0 && gm.state != S_INIT) return false;
    if (gm.state == S_ROOT_READY && !gm.nodes[(size_t)gm.root].has_prior) return false;       // add_noise mixes into the root's priors
    if (gm.state == S_CHOOSE && gm.nodes[(size_t)gm.root].n_kids == 0) return false;          // pick_move returns one of the root's children
    if (gm.pending_expand != -1 && gm.state != S_WAIT_ROOT && gm.state != S_WAIT_LEAF && gm.state != S_PLAYOUT) return false;
    gm.table_rebuild(gm.nodes.size());
    dst = std::move(gm);
    return true;
}
}  // namespace

extern "C" {

long bk_pool_snapshot(const bk_pool* p, int g, void* buf, long cap) {
    if (!p || g < 0 || g >= (int)p->games.size()) return -1;
    const Game& gm = p->games[g];
    if (gm.has_request() || gm.po >= 0 || gm.po_mark >= 0) return -2;      // between a deliver and the next collect only
    SnapWriter w;
    snapshot_game(gm, w);
    if (buf && cap >= (long)w.b.size()) std::memcpy(buf, w.b.data(), w.b.size());
    return (long)w.b.size();
}

int bk_pool_restore(bk_pool* p, int g, const void* buf, long len) {
    if (!p || g < 0 || g >= (int)p->games.size() || !buf || len <= 0) return -1;
    if (p->games[g].has_request()) return -2;
    for (int a : p->active)
        if (a == g) return -2;
    Game& dst = p->games[g];
    const int row_cap = p->row_cap > 0 ? p->row_cap : dst.row_cap;
    if (!restore_game(dst, static_cast<const uint8_t*>(buf), (size_t)len)) return -3;
    dst.row_cap = std::min(dst.row_cap, row_cap);                 // the receiving pool's collects may be smaller
    dst.prm.speculate_rows = std::min(dst.prm.speculate_rows, dst.row_cap);
    return 0;
}

/* ---- the step loop in C (selfplay.py:run_pools without the interpreter between two steps) ---- */
void bk_normalise_rows(float* probs, int n_rows) {
    for (int r = 0; r < n_rows; ++r) {
        float* row = probs + (size_t)81 * r;
        float sum = 0.f;
        for (int k = 0; k < 81; ++k) sum += row[k];                // left to right, fp32: the order is part of the contract
        for (int k = 0; k < 81; ++k) row[k] /= sum;
    }
}

int bk_pools_run(bk_pool* const* pools, int n_pools, const bk_evaluator* ev, int cap, bk_run_info* out) {
    if (!pools || n_pools <= 0 || n_pools > 16 || !ev || !ev->submit || !ev->wait || cap < 82) return -1;
    struct Lane {
        std::vector<bk_pos> recs;
        std::vector<float> probs, values;
        int64_t ticket = 0;
        int n = 0, npol = 0;
        bool live = true;
    };
    std::vector<Lane> lanes((size_t)n_pools);
    for (auto& l : lanes) {
        l.recs.resize((size_t)cap);
        l.probs.resize((size_t)cap * 81);
        l.values.resize((size_t)cap);
    }
    bk_run_info info{};
    const auto t0 = std::chrono::steady_clock::now();
    int rc = 0, out_now = 0;
    // wait for lane j's request and hand its rows to the pool.  A ticket that has been waited for is spent whatever the outcome: it
    // is cleared BEFORE the result is looked at, so that the clean-up below does not wait for it a second time (the engine would
    // answer "unknown ticket" and that message would replace the real cause in bk_last_error: ADVICE r5)
    auto finish = [&](int j) -> int {
        Lane& l = lanes[(size_t)j];
        const auto w0 = std::chrono::steady_clock::now();
        const int wrc = ev->wait(ev->ctx, l.ticket);
        info.wait_seconds += std::chrono::duration<double>(std::chrono::steady_clock::now() - w0).count();
        l.ticket = 0;
        --out_now;
        if (wrc) return wrc < 0 ? wrc : -wrc;
        bk_normalise_rows(l.probs.data(), l.npol);                  // Categorical(probs) re-normalises (nnet.py:274)
        bk_pool_deliver(pools[j], l.probs.data(), l.values.data());
        return 0;
    };
    for (bool busy = true; busy && !rc;) {
        busy = false;
        for (int i = 0; i < n_pools && !rc; ++i) {
            Lane& l = lanes[(size_t)i];
            if (l.ticket && (rc = finish(i))) break;
            if (l.live) {
                l.n = bk_pool_collect_pos(pools[i], l.recs.data(), cap, &l.npol);
                if (l.n == 0) { l.live = false; continue; }
                // never more than BK_POOLS_MAX_INFLIGHT requests out (the engine's evaluator has BK_MAX_INFLIGHT = 4 tickets): with
                // more pools than that the oldest request -- the next lane in turn that has one -- is taken in first
                for (int d = 1; out_now >= BK_POOLS_MAX_INFLIGHT && d < n_pools && !rc; ++d) {
                    const int j = (i + d) % n_pools;
                    if (lanes[(size_t)j].ticket) rc = finish(j);
                }
                if (rc) break;
                l.ticket = ev->submit(ev->ctx, l.recs.data(), l.n, l.npol, l.probs.data(), l.values.data());
                if (l.ticket <= 0) { rc = l.ticket < 0 ? (int)l.ticket : -1; l.ticket = 0; break; }
                ++out_now;
                info.steps += 1;
                info.rows += (uint64_t)l.n;
                info.policy_rows += (uint64_t)l.npol;
                busy = true;
            }
            busy = busy || l.ticket != 0;
        }
    }
    if (rc)                                   // requests still out: wait for them so that the evaluator's buffers can go
        for (auto& l : lanes)
            if (l.ticket) (void)ev->wait(ev->ctx, l.ticket);
    info.seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    if (out) *out = info;
    return rc;
}

}  // extern "C"
