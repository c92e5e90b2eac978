/*
 * bokego_tree.h -- C ABI of the native PUCT search core / lock-step game pool (libbkgo.so, host only).
 *
 * Replaces, for many concurrent games, the reference's per-tree Python loop
 *   MCTS.rollout / _descend / _puct_select / _expand / _backpropagate / choose / set_root
 *   (bokego/mcts.py:110-234) and the lazy node accessors Go_MCTS.dist / .value (mcts.py:371-403),
 * whose cache misses are exactly "enqueue a leaf".  The pool turns those misses into batches:
 *
 *   for (;;) {
 *       B = bk_pool_collect_pos(pool, recs, cap, &n_policy); // every game runs until it needs a network
 *       if (B == 0) break;                                   // all games finished
 *       t = bk_submit_positions(engine, recs, B, n_policy, PROBS|VALUE, NULL, probs, values);   // planes made on the GPU
 *       bk_wait(engine, t);
 *       ... renormalise probs rows as torch's Categorical does (nnet.py:274) ...
 *       bk_pool_deliver(pool, probs, values);
 *   }
 * (bk_pool_collect + bk_submit_prefix is the same loop with the planes encoded on the host.)  Several pools
 * rotate through one engine so that the host advances one while the GPU evaluates the others.
 */
#ifndef BOKEGO_TREE_H
#define BOKEGO_TREE_H

#include <stdint.h>

#include "bokego_go.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct bk_search_params {
    int32_t rollouts;      /* per move (BASELINE config 3: 1600, config 4: 400)                 */
    int32_t expand_thresh; /* MCTS kwarg expand_thresh, default 100 (mcts.py:61)                */
    double c_puct;         /* exploration_weight, default 4.0 (mcts.py:63)                      */
    float noise_weight;    /* Dirichlet(0.1) mix at every new root, default 0 (mcts.py:64)      */
    int32_t sample_plies;  /* self-play: first plies sampled ~ visit counts (0 = always argmax) */
    int32_t max_turns;     /* MAX_TURNS = 80 (mcts.py:13)                                       */
    int32_t eager;         /* 1: value of every new child is requested at expansion             */
    float komi;            /* 5.5 (go.py:45)                                                    */
    int32_t record_visits; /* 1: keep every ply's root-child visit counts (self-play records)    */
    int32_t prune;         /* 0: keep every node ever created, like the reference's never-pruned
                              Q/N/V dicts (a position from an abandoned branch that is reached again
                              keeps its statistics); 1: drop everything outside the new root's
                              subtree at each move (bounded memory for long self-play runs)      */
    int32_t speculate;     /* > 0: a leaf whose visit count reaches this value is marked "likely to be expanded";
                              the next evaluation request that goes out anyway also carries that node's policy
                              and its would-be children's values, so that its expansion -- after expand_thresh
                              visits -- needs no round trip of its own.  The networks are pure functions and
                              the tree is not touched: the search is the same search, rollout for rollout; only
                              n_value_evals / n_requests differ.  0: off                                     */
    int32_t speculate_rows; /* a request takes speculative rows only while it stays within this many rows
                              (default 128: the range of the engine's cooperative small-batch launch); clamped to
                              the `cap` of bk_pool_collect*, so that no request can outgrow a collect.  The
                              "same search" statement holds for networks whose outputs do not depend on the batch a
                              row travels in: the fp32 kernel always, the f16x2 kernel except for a request redone in
                              fp32 after an fp16-range overflow (the rows sharing that request change with it)     */
    int32_t request_tasks;  /* > 0: keep a request within this many network tasks (2 per policy row, 1 per value-only row)
                              where the search allows it: a candidate for evaluation ahead goes out in parts (its policy row and
                              as many children as fit, the rest -- by prior -- with later requests), and the children of a node
                              whose priors are known wait for a later request when an expansion would overflow.  64 is the range
                              in which the fp32 engine's cooperative launch gives a board 4 CUs (104 us per request; 140 us
                              from 65 tasks).  Values are pure functions of the position: the search is unchanged.  0: off  */
    int32_t eager_top;     /* > 0: instead of every new child (eager), request at an expansion only the values of the eager_top
                              children with the highest priors -- the ones the search is going to visit first: PUCT orders a
                              node's unvisited children by prior, and a 1600-rollout search ever visits 4 of a node's ~75
                              children in the median, 10 at the 90th percentile.  A node whose priors are not known yet sends
                              its policy row only; any other child is evaluated when a rollout first ends on it, together with
                              its eager_top - 1 next-best unevaluated siblings.  Same search (values are pure functions of the
                              position; the reference evaluates every value on first use, mcts.py:393-403), 5-7x fewer
                              evaluations than eager = 1.  0: off                                                          */
    int32_t request_steps[3]; /* the sizes at which a request gets dearer, ascending, [0] = request_tasks (fp32 engine: 64, 80,
                              128 -- 4 / 3 / 2 CUs per board): a request that is already beyond one step -- an expansion whose
                              priors are not known yet sends all its children -- takes passengers up to the next one     */
    int32_t branch_num;    /* MCTS kwarg branch_num (mcts.py:62,189-190, Go_MCTS.find_children(k), mcts.py:309-317): > 0 and < 81: a
                              node's children are the LEGAL moves among the branch_num moves with the highest prior (ties: lower
                              point first), so a node is expanded once its policy is known -- an expansion whose priors are still
                              out waits for them (one request) and happens before the next rollout.  0: every legal move         */
    int32_t simulate;      /* MCTS kwarg no_sim = False (boke.py --simulate; mcts.py:58,147-148,195-206): every rollout ends in a
                              playout from its leaf -- moves sampled from the policy (Go_MCTS.get_move, mcts.py:348-360, with
                              go.possible_eye), to a pass or turn > max_turns, scored by Game.score -- whose result (+1 / -1 for
                              the side to move at the leaf) is summed into Q along the path (mcts.py:208-214).  Draws come from
                              the game's own generator.  Where no acceptable move is left the playout passes (the reference
                              raises there; see tests/golden/simulate_playouts.json).  Playout positions outside the tree are
                              evaluated (policy) and forgotten.  Turns evaluation ahead (speculate) off.  0: off            */
    int32_t use_value;     /* 0: there is no value net (MCTS(value_net=None), mcts.py:68-69): no value is ever asked for, V stays
                              0; rows that carry one anyway are ignored.  Needs simulate.  1: default                       */
    double value_weight;   /* MCTS kwarg value_net_weight (mcts.py:65-72): a child's average reward in the selection is
                              ((1 - w) Q + w V) / N.  1.0 without simulation (the reference forces it), else default 0.5, 0
                              without a value net                                                                            */
    int32_t leaves;        /* OPT-IN THROUGHPUT MODE, NOT the reference's search ("virtual loss only as an opt-in
                              throughput mode"; default 1 = off, every parity statement in this header is about 1).  > 1: a search
                              step gathers up to `leaves` rollouts that wait for a value before it sends its request: each
                              waiting rollout leaves a VIRTUAL LOSS on its path (every node below the root counts one more
                              visit, lost by the side that moved there), so that the next descent goes elsewhere; when the values
                              are back the losses are taken off and the rollouts are backed up in the order they were made.
                              Rollouts whose leaf value is already known are backed up at once, as ever.  A request then carries
                              ~`leaves` times the rows -- a 64-game pool asks for a full round of workgroups instead of a
                              third of one -- and a move needs ~1 / leaves of the round trips.  Deterministic per game (the
                              games still do not depend on sharding, threads or batch grouping), every rollout is backed up
                              exactly once (the visit totals of a search are those of `leaves` = 1), but the tree it builds is
                              another one: rollouts descend on statistics that are up to leaves - 1 backups behind.
                              Ignored (= 1) with simulate, branch_num or without a value net; turns speculate off.
                              Measured against the one-leaf search (100 games, profiles/r06_leaves_probe.txt): 8 leaves lose
                              34 : 66 at 400 rollouts per move and win 61 : 39 at 1600; see leaves_visit_only.                  */
    int32_t leaves_visit_only; /* multi-leaf mode: 1 = a waiting rollout leaves a virtual VISIT on its path (N + 1) instead of a virtual loss
                                  (N + 1, V + 1): a milder push away from the path -- 47 : 53 instead of 34 : 66 at 8 leaves and 400
                                  rollouts per move, for about two thirds of the throughput gain (64 games per rank 0.168 ->
                                  0.134 s at 16 leaves against 0.123).  Default 0                                                 */
} bk_search_params;

typedef struct bk_game_info {
    int32_t done;
    int32_t n_moves;
    float score;           /* Tromp-Taylor area score of the final position, black - white - komi */
    int32_t n_nodes;
    uint64_t n_value_evals, n_policy_evals, n_requests;
    int32_t root_N;
    int32_t reserved;
    double root_V;
} bk_game_info;

typedef struct bk_pool bk_pool;

void bk_search_params_default(bk_search_params *p);
/* seeds[i] drives game i's noise / move sampling: results do not depend on how games are grouped */
bk_pool *bk_pool_create(int n_games, const bk_search_params *prm, const uint64_t *seeds, int threads);
void bk_pool_destroy(bk_pool *p);

/* Advance every unfinished game until it needs network outputs; write the requested positions'
 * feature planes (uint8 [B][27][9][9]) to feats: first the n_policy positions that need policy
 * (+value), then the value-only ones.  Returns B (0 when every game is over).  cap >= 82. */
int bk_pool_collect(bk_pool *p, uint8_t *feats, int cap, int *n_policy);
/* Same batch, as 192-byte position records instead of planes: each requested node's liberty cache is
 * refreshed in place (the history-dependent half of nnet.features(), go.py:220-243) and the record copied
 * to out[B]; the planes are a pure function of the record (bk_pos_features_u8 on it gives exactly what
 * bk_pool_collect would have written) and are computed on the GPU by bk_submit_positions. */
int bk_pool_collect_pos(bk_pool *p, bk_pos *out, int cap, int *n_policy);
/* Soft limit of a batch in network tasks (2 per policy row, 1 per value-only row; 0 = none): the batch stops taking games'
 * requests where one more would push it over -- e.g. 768 = one round of 3-board workgroups on 256 CUs, beyond which the fp32
 * engine's launch takes 1.1 instead of 0.75 ms.  A game left out keeps its request and goes first into the next batch; what a
 * game computes does not depend on the batch it travels in. */
void bk_pool_set_task_cap(bk_pool *p, int tasks);
/* probs: [n_policy][81] (already Categorical-normalised), values: [B], same order as collected */
void bk_pool_deliver(bk_pool *p, const float *probs, const float *values);

/* host seconds this pool has spent so far advancing its games (select / expand / backup), writing request rows, and taking
 * deliveries: out3 = {advance, emit, deliver} (tools/selfplay_breakdown.py) */
void bk_pool_phase_seconds(const bk_pool *p, double *out3);
/* In-batch de-duplication for bk_pool_collect_pos (off by default): request rows of one batch whose 192-byte records are equal
 * byte for byte -- games still in the same opening -- travel once and every asker gets the answer at deliver.  The reference
 * keeps one memo for all trees (mcts.py:41-44); here a game has its own, and this is the part of the shared one that costs
 * nothing to keep exact: the networks' outputs for equal records are the same bits.  Per-game counters do not change. */
void bk_pool_set_dedup(bk_pool *p, int on);
/* Worker-thread affinity of the games (on by default): game g is advanced and delivered to by the same worker step after step, so
 * its tree stays in that core's caches; 0 = items handed out one by one from a single counter, deliveries on the calling thread
 * (what the environment variable BK_NO_LANES selected until round 5: no request path reads the environment any more).  Which
 * thread works on a game never changes what the game computes. */
void bk_pool_set_lanes(bk_pool *p, int on);
void bk_pool_dedup_rows(const bk_pool *p, uint64_t *requested, uint64_t *sent);
/* stress of the worker threads the pools share (`jobs` short parallel regions on `threads` threads); 0 = every item ran once */
int bk_team_selftest(int threads, int jobs);
/* two callers at once, each with a region that waits (at most timeout_ms) to see the other's running: 0 = they ran side by side */
int bk_team_selftest_concurrent(int threads, int timeout_ms);
int bk_pool_n_games(const bk_pool *p);
int bk_pool_n_done(const bk_pool *p);
int bk_pool_game_info(const bk_pool *p, int g, bk_game_info *out);
/* Visit / value statistics of the moves game g has chosen so far -- what the end-of-generation all-reduce sums over all games
 * (north_star: "all-reduce visit/value statistics at the end of a generation"; the reference's workers append their results to a
 * Manager().list(), bin/selfplay.py:179-180,201-204):
 *   root_visits[m]      sum over the plies played of N[child reached by move m] of the root at the moment the ply was chosen
 *                       (MCTS.choose reads the same counts, mcts.py:122-128)
 *   sum_root_value      sum over those plies of V[root] / N[root] (= 2 MCTS.winrate() - 1, mcts.py:159-170: the side to move's view)
 *   sum_abs_root_value  ... of its magnitude;  n_root_values: the number of plies in the two sums
 * The two sums are kept in 2^-32 fixed point, so totals over games are exact and do not depend on the order of addition. */
typedef struct bk_game_stats {
    uint64_t root_visits[81];
    double sum_root_value, sum_abs_root_value;
    uint64_t n_root_values;
} bk_game_stats;
int bk_pool_game_stats(const bk_pool *p, int g, bk_game_stats *out);
int bk_pool_game_moves(const bk_pool *p, int g, int16_t *out, int cap);
/* visit counts of the root's children when ply `ply` was chosen (needs record_visits); returns their number */
int bk_pool_game_visits(const bk_pool *p, int g, int ply, int16_t *moves, int32_t *N);
/*
 * Manual control: the reference's single-tree surface on a pool game.  With manual mode on, a game
 * only searches while it has rollouts outstanding and never moves by itself:
 *   bk_pool_add_rollouts  = MCTS.rollout(n)        (mcts.py:133-151; drive collect/deliver until B == 0)
 *   bk_pool_choose        = MCTS.choose()          (mcts.py:110-131; re-roots, returns the move)
 *   bk_pool_play          = set_root(root.make_move(mv))  (gtp.py:332-337: an outside move)
 *   bk_pool_set_position  = set_root(Go_MCTS(board=...))  (clear_board, handicap)
 */
void bk_pool_set_manual(bk_pool *p, int on);
/* change bk_search_params.speculate / speculate_rows / request_tasks of every game (e.g. after the engine's precision was switched) */
void bk_pool_set_speculation(bk_pool *p, int speculate, int rows, int request_tasks);
int bk_pool_add_rollouts(bk_pool *p, int g, int n);
int bk_pool_choose(bk_pool *p, int g);
int bk_pool_play(bk_pool *p, int g, int move);
int bk_pool_set_position(bk_pool *p, int g, const bk_pos *pos);
int bk_pool_root_pos(const bk_pool *p, int g, bk_pos *out);
/* root children of game g in ascending move order; returns their number */
int bk_pool_root_children(const bk_pool *p, int g, int16_t *moves, int32_t *N, double *V);

/*
 * Read-only views of a game's tree.  The reference keeps Q / N / V / children as dicts keyed by node that anybody may
 * read (mcts.py:46-52; GTP.analyze reads self.N[n], gtp.py:386,395); here a node is an id, found by its position:
 *   bk_pool_find            id of the node with this (board, ko, last move, side), -1 if the tree never saw it
 *   bk_pool_node            statistics (+ the 192-byte position) of node `id`
 *   bk_pool_node_children   ids of its children (as linked by expansion); returns their number
 *   bk_pool_node_prior      its 81 move priors (the Categorical-normalised policy, noise included); -1 if not evaluated
 *   bk_pool_principal_variation   the moves of the most visited line from the root
 * Ids stay valid until the next re-rooting of a pruning tree (prm.prune).
 * bk_pool_set_analyze(on) + bk_pool_variation: MCTS.rollout(n, analyze_dict) (mcts.py:143-147) -- while on, every
 * descent longer than two nodes is remembered under the root child it passed through; bk_pool_variation copies the node
 * ids of the remembered line (the child first) for the root child reached by `move` and returns its length (0: none).
 * Switching analysis on or off, and every re-rooting, forgets the remembered lines.
 */
#define BK_NODE_EXPANDED 1
#define BK_NODE_TERMINAL 2
#define BK_NODE_HAS_VALUE 4
#define BK_NODE_HAS_PRIOR 8
typedef struct bk_node_info {
    int32_t N;          /* visits (MCTS.N[node])                                   */
    int32_t n_children;
    double V;           /* summed backed-up values (MCTS.V[node])                  */
    float value;        /* the ValueNet's output for this position (Go_MCTS.value) */
    int16_t move;       /* the move that leads here (last_move)                    */
    uint16_t flags;     /* BK_NODE_*                                               */
} bk_node_info;
int bk_pool_find(const bk_pool *p, int g, const bk_pos *pos);
int bk_pool_root_id(const bk_pool *p, int g);
int bk_pool_node(const bk_pool *p, int g, int id, bk_node_info *out, bk_pos *pos);
int bk_pool_node_q(const bk_pool *p, int g, int id, double *q);    /* MCTS.Q[node]: summed playout rewards (simulate) */
int bk_pool_node_children(const bk_pool *p, int g, int id, int32_t *ids, int cap);
int bk_pool_node_prior(const bk_pool *p, int g, int id, double *prior);
int bk_pool_principal_variation(const bk_pool *p, int g, int16_t *moves, int cap);
void bk_pool_set_analyze(bk_pool *p, int on);
int bk_pool_variation(const bk_pool *p, int g, int move, int32_t *ids, int cap);

/*
 * Snapshot / restore of one game's search state (replaces: MCTS.__getstate__ / __setstate__ / __deepcopy__, mcts.py:81-108, and
 * Go_MCTS.__getstate__ / __setstate__, mcts.py:281-292 -- the reference pickles a tree without its nets and deep-copies one that
 * shares them): nodes, edges, priors, N / V / Q, root, moves played, visit records, counters, the game's generator state and
 * its search parameters; the networks are not part of it.  A restored game continues rollout for rollout as the original would.
 *   bk_pool_snapshot   returns the snapshot's size in bytes; the bytes are written when buf != NULL and cap is large enough
 *                      (call with NULL first).  -1: bad argument; -2: the game has a request out (snapshots are taken between a
 *                      deliver and the next collect -- for a manually driven game: whenever the caller holds it)
 *   bk_pool_restore    replaces game g of ANY pool by the snapshot (its search parameters come with it).  0, -1 bad argument,
 *                      -2 game g has a request out, -3 not a snapshot of this build (the game is left as it was)
 * The format is private to one build of the library on one host architecture (it carries its structure sizes).
 */
long bk_pool_snapshot(const bk_pool *p, int g, void *buf, long cap);
int bk_pool_restore(bk_pool *p, int g, const void *buf, long len);

/*
 * The whole step loop in C: what bokego_amd/selfplay.py:run_pools and examples/bk_selfplay.c do -- for every pool in turn: wait
 * for its batch, re-normalise the priors (bk_normalise_rows), deliver, collect the next batch (position records), submit -- until
 * every game of every pool is over; several pools rotate so that the host advances one while the evaluator works on the
 * others' batches.  Between two steps there is no interpreter: a 64-game pool's step is a 100-170 us kernel, and 130 us of
 * Python per step were what kept three small pools from overlapping (profiles/r04_pools_probe.txt).
 * The evaluator is two callbacks: submit() starts the evaluation of B records (the first n_policy need policy + value, the rest
 * the value only) and returns a ticket > 0 (or a negative error); probs[n_policy][81] (softmax outputs, not yet re-normalised)
 * and values[B] must be filled in when wait(ticket) returns 0.  bk_engine_evaluator() (include/bokego_amd.h) fills one in for
 * a HIP engine: bk_submit_positions / bk_wait.
 * Returns 0, -1 for bad arguments, or the evaluator's error code (outstanding tickets are waited for first; the pools are
 * then in the middle of a step and cannot be driven on).
 */
#define BK_POOLS_MAX_INFLIGHT 4 /* bk_pools_run keeps at most this many requests out at once (= the engine's BK_MAX_INFLIGHT) */
typedef struct bk_evaluator {
    void *ctx;
    int64_t (*submit)(void *ctx, const bk_pos *recs, int B, int n_policy, float *probs, float *values);
    int (*wait)(void *ctx, int64_t ticket);
} bk_evaluator;
typedef struct bk_run_info {
    uint64_t steps, rows, policy_rows;   /* batches submitted, rows in them, of which policy rows */
    double seconds, wait_seconds;        /* wall time of the call; of which blocked in wait()      */
} bk_run_info;
int bk_pools_run(bk_pool *const *pools, int n_pools, const bk_evaluator *ev, int cap, bk_run_info *out);
/* p / sum(p) per row of 81, as torch's Categorical(probs) does (nnet.py:274); the sum is taken left to right in fp32 */
void bk_normalise_rows(float *probs, int n_rows);

#ifdef __cplusplus
}
#endif
#endif
