/*
 * bokego_comm.h -- C ABI of the one collective on the path: the end-of-generation reduction of self-play
 * statistics over the GPUs of a node (libbkcomm.so = RCCL over xGMI behind plain pointers).
 *
 * The reference has no collective: its workers append to a multiprocessing Manager().list()
 * (bin/selfplay.py:179-180,201-204) and the parent sums.  Here every rank (one process per GPU) plays its
 * shard of the games (game id % world) with no communication, then ONE all-reduce(sum) of a short fp64
 * vector -- 173 values, 1,384 bytes: [games, black wins, white wins, plies, sum of scores, value evals, policy evals,
 * requests, sum / sum of magnitudes / count of the root's mean backed-up value at every move, first-move histogram[81],
 * histogram of root-child visit counts over every move[81]] (bokego_amd/selfplay.py:STATS_FIELDS; the per-game figures come
 * from bk_pool_game_stats, include/bokego_tree.h; every entry is an integer or a multiple of 2^-32, so the sums are exact
 * and independent of the reduction order) -- gives every rank the generation's visit / value statistics.  The Python host uses
 * torch.distributed (backend "nccl" = the same RCCL) for this; libbkcomm.so is the same step for hosts
 * without Python/torch.
 *
 * Bootstrap: rank 0 calls bk_comm_unique_id and hands the 128 bytes to the other ranks by whatever the
 * launcher offers (file, socket, environment); then every rank calls bk_comm_init.
 */
#ifndef BOKEGO_COMM_H
#define BOKEGO_COMM_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define BK_COMM_ID_BYTES 128

typedef struct bk_comm bk_comm;

int bk_comm_abi_version(void);
/* rank 0: create the rendezvous id (ncclGetUniqueId) */
int bk_comm_unique_id(uint8_t id[BK_COMM_ID_BYTES]);
/* every rank: join; device_id = the GPU this process drives (normally the local rank) */
int bk_comm_init(int rank, int world, const uint8_t id[BK_COMM_ID_BYTES], int device_id, bk_comm **out);
/* in-place sum over all ranks of a host fp64 vector (n <= 4096): one ncclAllReduce on the comm's stream */
int bk_comm_allreduce_sum_f64(bk_comm *c, double *buf, int n);
/* generation start (optional): rank `root`'s n fp32 values -- a net's tensors back to back, 2 x 3.9 MB for
 * both nets -- to every rank, in place: one ncclBroadcast through a device buffer grown on demand.  The reference
 * shares weights between its worker processes with share_memory() (bin/selfplay.py:171-175). */
int bk_comm_broadcast_f32(bk_comm *c, float *buf, int64_t n, int root);
/* every rank waits here until all have arrived (a one-word all-reduce): called in front of the statistics' all-reduce so that its
 * time can be told from the wait for the slowest rank (bench.py: stats_allreduce_ms = the collective, allreduce_wait_ms = the skew) */
int bk_comm_barrier(bk_comm *c);
/* what the first multi-GPU line should say about its fabric: RCCL's version (ncclGetVersion: e.g. 22205) and the PCI address of
 * the GPU this communicator drives ("0000:05:00.0"; cap >= 16) */
int bk_comm_rccl_version(void);
int bk_comm_device_pci(const bk_comm *c, char *out, int cap);
int bk_comm_rank(const bk_comm *c);
int bk_comm_world(const bk_comm *c);
int bk_comm_destroy(bk_comm *c);
/* message of the last failure on this thread; 0 = ok, negative = failure as in bokego_amd.h */
const char *bk_comm_last_error(void);

#ifdef __cplusplus
}
#endif
#endif
