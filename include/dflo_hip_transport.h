/*
 * dflo_hip_transport.h -- the seams BELOW the contract of dflo_hip.h: what the native multi-device driver
 * (dflo_amd/csrc/multi.hip, dflo_hip_multi_*) is written against -- the stage split by shard set, pack / unpack of halo records,
 * delivery by the kernels themselves, sequence words, the time-step table -- and inspection of a multi-device handle.
 * A host program needs this header only to bring a transport of its own below dflo_hip_multi_create_rank_custom's level
 * (INTEGRATION.md section 4); everything here replaces LA::distributed::Vector::update_ghost_values / Utilities::MPI::min of
 * the MPI variant (src_mpi/claw.cc:793, 579; src_mpi/limiter.cc:232).
 */
#ifndef DFLO_HIP_TRANSPORT_H
#define DFLO_HIP_TRANSPORT_H

#include "dflo_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

/* Launch all work of this handle on an existing HIP stream (hipStream_t passed
 * as void*); NULL = the handle's own stream. */
int dflo_hip_set_stream(dflo_hip_handle h, void *hip_stream);
int32_t dflo_hip_dofs_per_cell(dflo_hip_handle h);

/* ------------------------------------------------ multi-device halo seam */
/* Replaces LA::distributed::Vector::update_ghost_values() of the MPI variant
 * (src_mpi/claw.cc:793, src_mpi/limiter.cc:232).  The engine owns cells
 * [0,n_owned) and reads ghost cells [n_owned,n_cells).  pack gathers the DoFs
 * ([n][ndof]) or the cell averages ([n][4]) of the listed owned cells into a
 * contiguous device buffer; unpack scatters a received buffer ([n_ghost][ndof] or
 * [n_ghost][4], ghost order) into the ghost cells.  The transport between the
 * two (RCCL send/recv through torch.distributed) is the caller's.
 * With a TVB limiter the ghost AVERAGES must be refreshed between the update and
 * the limiter of a stage (the MPI variant computes cell averages on owned+ghost
 * cells after the first update_ghost_values, src_mpi/claw.cc:793,653-669):
 *   dflo_hip_stage_update -> exchange averages -> dflo_hip_stage_limit -> exchange DoFs.
 * dflo_hip_stage == dflo_hip_stage_update + dflo_hip_stage_limit. */
int dflo_hip_stage_update(dflo_hip_handle h, int rk, double dt);
int dflo_hip_stage_limit(dflo_hip_handle h);
/* The same stage split by shard set, for overlapping the exchange with compute (what dflo_amd/csrc/multi.hip is written
 * against): part 1 = rim shards (those that read ghost cells), part 2 = interior shards, part 0 = all; for the update of a
 * stage that a TVB limiter follows also part 3 = rim shards + the ring of shards next to them (the limiter of a rim cell
 * reads the new averages of its neighbours there) and part 4 = the others.  dflo_hip_set_stream chooses the stream of
 * the following launches; launches of different parts of one stage may run side by side on different streams (they read
 * the previous stage and write disjoint shards); ordering between the streams is the caller's (events).
 *   open -> update_part(1) -> [limit_part(1) -> pack -> exchange -> unpack on a second stream]
 *        -> update_part(2) -> limit_part(2) -> finish (reductions, CFL minimum)                    */
int dflo_hip_stage_open(dflo_hip_handle h, int rk, double dt);
int dflo_hip_stage_update_part(dflo_hip_handle h, int part);
int dflo_hip_stage_limit_part(dflo_hip_handle h, int part);
int dflo_hip_stage_finish(dflo_hip_handle h);
int dflo_hip_n_rim_shards(dflo_hip_handle h);
int dflo_hip_n_part_shards(dflo_hip_handle h, int part);   /* shards of stage_update_part's part 0..4 (0: all) */
int dflo_hip_n_ghost_cells(dflo_hip_handle h);
int dflo_hip_set_send_cells(dflo_hip_handle h, int32_t n, const int32_t *cells);
int dflo_hip_pack_send(dflo_hip_handle h, void *device_buffer);
int dflo_hip_pack_send_avg(dflo_hip_handle h, void *device_buffer);
int dflo_hip_unpack_ghost(dflo_hip_handle h, const void *device_buffer);     /* also recomputes ghost averages */
int dflo_hip_unpack_ghost_avg(dflo_hip_handle h, const void *device_buffer);
/* Instead of unpack_ghost_avg: the limiter passes that follow (Qk) read the ghost cells' averages straight from the received
 * buffer ([n_ghost][4], ghost order) -- one small kernel less between the arrival of the averages and the limiter of the rim
 * cells, the stretch of a TVB stage that every neighbour waits for.  NULL, or the next dflo_hip_unpack_ghost_avg /
 * dflo_hip_unpack_ghost_cells / dflo_hip_set_solution, returns to the averages held by the engine.  Not for runs whose stage
 * kernels read ghost averages too (LxF flux). */
int dflo_hip_ghost_avg_source(dflo_hip_handle h, const void *device_buffer);
/* DoFs and cell average of every listed cell in one record, [n][ndof + 4]: the ghost copy then holds the bits of its owner
 * (an average formed again from the DoFs differs from the stage kernel's in the last place; the LxF flux and the TVB
 * differences read it).  What the native multi-device driver ships. */
int dflo_hip_pack_send_cells(dflo_hip_handle h, void *device_buffer);
int dflo_hip_unpack_ghost_cells(dflo_hip_handle h, const void *device_buffer);
/* ONE exchange per stage where a TVB limiter sits between the update and update_ghost_values (Qk, ghost cells known by their
 * traces, no KXRCF indicator) -- the reference's two (src_mpi/limiter.cc:232: the averages the limiter reads; src_mpi/claw.cc:793:
 * the limited state) merged.  The cut cells leave UNLIMITED, each with what its owner's limiter will read of its surroundings:
 * records [n][ndof + 20] = DoFs, the cell's average, the averages of its four face neighbours as the owner holds them (zeros where
 * there is none).  The receiver hands them to limit_ghost_cells: the NEXT dflo_hip_stage_limit_part(h, 1) -- the limiter pass over
 * its rim shards -- takes them along.  An unpack kernel puts them into the ghost shards; the pass runs over the rim shards and the
 * ghost shards -- the routine that limits the owned cells; a ghost cell's neighbour that is one of the receiver's cells is read
 * from the receiver's averages, one that lives with the ghost's owner from the record: the owner's inputs, arithmetic and bits --
 * and the ghost shards' wavefronts form the traces of the limited ghost cells into trace table `table` (0 | 1) themselves.  Sound
 * where every cut cell borders on ONE other part (the driver checks). */
int dflo_hip_pack_send_cells_unlimited(dflo_hip_handle h, void *device_buffer);
int dflo_hip_limit_ghost_cells(dflo_hip_handle h, const void *records, int table);
/* Face-trace records: when nothing needs more of a ghost cell than its trace on the cut faces and its average
 * -- Qk without the KXRCF indicator: dflo_hip_halo_traces() = 1 -- the stage kernels read the ghost cells from a table of
 * traces, [n_ghost_traces][4][k+1] doubles ordered by (ghost cell, face), and the halo message of a cut face shrinks from
 * the cell's (k+1)^2 * 4 doubles to (k+1) * 4 (Q2: 36 -> 12; the 4-double average travels with pack_send_avg).  The
 * sender lists its (owned cell, face) pairs in the receiver's order (set_send_faces) and packs their traces; the receiver
 * lets the transport write straight into one of the engine's two trace tables (ghost_trace_buffer) and switches the
 * stage kernels to it before the next stage (use_ghost_traces) -- no unpack kernel.  dflo_hip_set_solution fills both
 * tables from the ghost cells' DoFs.  DFLO_HALO_CELLS=1 keeps whole-cell records. */
int dflo_hip_halo_traces(dflo_hip_handle h);
int dflo_hip_n_ghost_traces(dflo_hip_handle h);
int dflo_hip_set_send_faces(dflo_hip_handle h, int32_t n, const int32_t *cells, const int32_t *faces);
int dflo_hip_pack_send_traces(dflo_hip_handle h, void *device_buffer);
int dflo_hip_ghost_trace_buffer(dflo_hip_handle h, int which, void **device_ptr);
int dflo_hip_use_ghost_traces(dflo_hip_handle h, int which);
/* The engine's two ghost-trace tables ([n_ghost_traces][4][k+1] doubles each) and its table of the parts' time-step minima
 * ([2][16] doubles) in memory of the CALLER's -- a window it exports to other processes as one allocation (the runtime serves small
 * allocations as fragments of shared blocks, which cannot be exported reliably one by one).  The current contents move along;
 * the buffers stay the caller's and must outlive the engine. */
int dflo_hip_set_ghost_trace_buffers(dflo_hip_handle h, void *table0, void *table1);
int dflo_hip_set_dt_table_buffer(dflo_hip_handle h, void *table);
/* The next stage or limiter kernel this engine launches (stage_update_part / stage_limit_part) carries `event` (a hipEvent_t)
 * as its completion signal -- hipExtLaunchKernel's stopEvent -- instead of the caller recording the event behind it: one packet
 * less between two kernels of a stream (the multi-device schedule orders its two streams with one such event per phase).
 * If that launch turns out to be empty the event is recorded the plain way. */
int dflo_hip_attach_event(dflo_hip_handle h, void *event);
/* 1 if the last dflo_hip_stage_finish put a kernel of its own on the engine's stream BEHIND the stage's last stage / limiter launch (boundary
 * programs no pass took along, the time step, the reductions): an event attached to that last launch then does not cover everything the
 * next stage reads, and a caller that orders a second stream by it has to record the plain way. */
int dflo_hip_finish_enqueued(dflo_hip_handle h);
/* Delivery by the stage kernel itself (one process per GPU over mapped tables; Qk, ghost cells by their traces).  set_deliver,
 * once per receive area (0 | 1): the records of the send list of set_send_faces go, segment by segment as in pack_send_to, to
 * dst[i] -- the neighbours' trace tables of that area --, and flags[i] are the neighbours' sequence words.  stage_deliver arms
 * the NEXT launch over all shards (stage_update_part(h, 0) / stage(h, ..)): every workgroup whose shard has cut faces forms the
 * traces of its new state on them (the bits face_trace / pack_send_traces would give) and stores them at their destination;
 * the last such workgroup publishes `seq` in the words.  No rim launch of its own, no pack kernel, no second stream:
 * update_ghost_values (src_mpi/claw.cc:793) is part of the kernel that produced the values. */
int dflo_hip_set_deliver(dflo_hip_handle h, int area, int n_segments, const int32_t *first, void *const *dst, void *const *flags);
int dflo_hip_stage_deliver(dflo_hip_handle h, int area, uint64_t seq);
/* ... and the arrival of the neighbours' traces of the stage BEFORE can be awaited inside that launch as well: set_arrival_words
 * names this engine's own sequence words (one per neighbour that sends; fine-grained memory) and a host-mapped failure word;
 * stage_await(seq), together with stage_deliver, makes the workgroups of the shards that read ghost traces poll the words behind
 * their own loads until they have reached seq (DFLO_IPC_TIMEOUT_S, default 120 s: then the failure word goes up and the workgroup leaves the kernel without computing or delivering).  The other workgroups wait for nothing.  Only where
 * the trace tables are fine-grained memory (or written by this device itself): the traces are read inside the running kernel. */
int dflo_hip_set_arrival_words(dflo_hip_handle h, int n, void *const *words, void *fail);
int dflo_hip_stage_await(dflo_hip_handle h, uint64_t seq);
/* The next stage launch (Qk) does not END before the 64-bit word `word` (fine-grained memory) has reached seq: its first workgroup
 * polls for it once its own work is done (the time-out and failure word of set_arrival_words apply).  How the multi-device schedule
 * orders the compute stream's NEXT kernel behind the rim launch of the comm stream -- a one-thread kernel behind the rim launch
 * publishes the word -- without a wait packet in front of that next kernel (8.4 us of the compute stream even when long satisfied). */
int dflo_hip_stage_tail_wait(dflo_hip_handle h, const void *word, uint64_t seq);
/* ... and who publishes that word at no cost: the next pack kernel this engine launches (pack_send*, pack_send_to*: the first kernel of
 * the comm stream behind the rim launch, whose stores the kernel boundary has released) stores seq into `word` with its first thread,
 * before anything else. */
int dflo_hip_pack_publish(dflo_hip_handle h, void *word, uint64_t seq);
/* With a TVB limiter between update and send (src_mpi/limiter.cc:232: a second update_ghost_values) the exchange rides in BOTH
 * kernels of a stage.  set_deliver_averages, once per receive area: the averages of the cells of set_send_cells go to dst[i] (the
 * neighbours' average areas: [4] doubles per cell), flags[i] are the neighbours' words for them, words[] this engine's own words
 * for the neighbours' averages.  stage_deliver_averages arms the next launch over all shards: the workgroups of the shards on a
 * cut deliver their cells' new averages (and the stage kernel keeps those shards off the list of marked shards).  limit_exchange
 * arms the next limiter pass over all shards (stage_limit): one extra wavefront per shard on a cut waits for the neighbours'
 * averages to reach average_seq (poll_in_kernel; else the caller has waited), limits the shard with them (ghost_avg_source) and
 * delivers the traces of the limited state into the neighbours' tables of trace_area, publishing trace_seq.  Needs a pass that
 * walks the list of marked shards (limiter_walks_list: TVB on Qk squares with marks). */
int dflo_hip_set_deliver_averages(dflo_hip_handle h, int area, int n_segments, const int32_t *first, void *const *dst, void *const *flags,
                                  int n_words, void *const *words, void *fail);
int dflo_hip_stage_deliver_averages(dflo_hip_handle h, int area, uint64_t seq);
int dflo_hip_limit_exchange(dflo_hip_handle h, int trace_area, uint64_t trace_seq, uint64_t average_seq, int poll_in_kernel);
int dflo_hip_limiter_walks_list(dflo_hip_handle h);
/* The kernels that deliver store their values at system scope -- written through where the destination is fine-grained memory
 * -- and wait for them; where the destinations are PLAIN device memory (of another process on this device: only a release writes
 * such stores back) every delivering workgroup also has to fence: plain = 1. */
int dflo_hip_deliver_to_plain_memory(dflo_hip_handle h, int plain);
/* Pack and deliver in one kernel (several engines in one process): records [first[i], first[i+1]) of the send list are
 * written at dst[i] -- the receive area of the i-th peer, on this device or on another one reached over xGMI peer access --
 * instead of into a staging buffer that a copy per peer then moves.  kind: 0 whole cells ([ndof + 4] doubles per record,
 * as pack_send_cells), 1 cell averages ([4], as pack_send_avg), 2 face traces ([4 (k+1)], as pack_send_traces; the send
 * list is that of set_send_faces), 3 unlimited cells with their neighbours' averages ([ndof + 20], as
 * pack_send_cells_unlimited).  n_segments <= 16. */
int dflo_hip_pack_send_to(dflo_hip_handle h, int kind, int n_segments, const int32_t *first, void *const *dst);
/* The same, and the kernel tells the receivers: once every record of the launch is visible system-wide, the workgroup that
 * finishes last stores `seq` (release, system scope) into the 64-bit words flags[i] -- sequence words in the receivers'
 * fine-grained memory, which a one-wavefront wait kernel on the receiver's comm stream polls.  One process per GPU without a
 * transport library on the per-stage path: the receive areas and the words are mapped through hipIpcGetMemHandle /
 * hipIpcOpenMemHandle once, at create (dflo_hip_multi_create_rank with DFLO_RANK_TRANSPORT=ipc).  Replaces the same
 * update_ghost_values (src_mpi/claw.cc:793, src_mpi/limiter.cc:232).  flags NULL: dflo_hip_pack_send_to. */
int dflo_hip_pack_send_to_signal(dflo_hip_handle h, int kind, int n_segments, const int32_t *first, void *const *dst,
                                 void *const *flags, uint64_t seq);
/* The time step of a run over several engines (Utilities::MPI::min(global_dt), src_mpi/claw.cc:579) without a host hop and
 * without a kernel of its own.  Every engine keeps a device table mins[2][16] of raw CFL minima: row p holds, slot by slot,
 * the minima of all parts for the step of parity p (steps counted since set_solution).  The reductions that end a step write
 * this engine's minimum into slot my_slot of the next step's row -- of its own table and of the peer_tables handed over here
 * (engines of the same process: plain stores over xGMI peer access; entry my_slot and null entries are skipped) -- and every
 * consumer of the time step (stage kernels, boundary programs, the clock) takes the minimum over the n_slots of its row and
 * applies the rules of src/claw.cc:468-476 itself.  The caller orders the streams: the next step's first kernel after the
 * peers' reductions.  One process per GPU: n_slots = 1 and an all-reduce(min) in place on dflo_hip_dt_slot (the slot the
 * next step to run reads) between the two.  n_slots = 0 (the default): one engine, the reductions apply the rules. */
int dflo_hip_dt_table(dflo_hip_handle h, void **table);
int dflo_hip_dt_exchange(dflo_hip_handle h, int my_slot, int n_slots, void *const *peer_tables);
int dflo_hip_dt_slot(dflo_hip_handle h, void **slot);

/* ------------------------------------------------ the multi-device handle: self-halo, inspection, per-part data */
/* Self-halo: ONE part on ONE device that is its own neighbour across a virtual cut (dflo_mesh_partition_self below:
 * n_virtual = 1 cuts at the periodic faces in x, >= 2 between the virtual parts of `partitioner`), driven through the complete
 * stage schedule of a multi-device run -- rim shards beside the interior on two streams, pack, transport into the trace table,
 * the time-step reduction -- where a plain one-part handle issues the single engine's launches.  A measuring device for boxes
 * with one GPU: its rate over the plain engine's bounds the weak-scaling efficiency of a rank whose neighbours are as fast as
 * itself (update_ghost_values / Utilities::MPI::min of src_mpi/claw.cc:793, 579 and src_mpi/limiter.cc:232 all happen, against
 * itself).  Results are those of the single engine, bit for bit on the nodal basis.  transport: dflo_self_transport --
 * DIRECT the one-process schedule (pack kernels store into the own trace table), RCCL the one-process-per-GPU schedule on a
 * one-rank communicator (grouped ncclSend / ncclRecv to itself, ncclAllReduce(min)), COPY staging buffer + hipMemcpyPeerAsync,
 * IPC the one-process-per-GPU schedule with the sequence-word transport of DFLO_RANK_TRANSPORT=ipc against itself. */
typedef enum { DFLO_SELF_DIRECT = 0, DFLO_SELF_RCCL = 1, DFLO_SELF_COPY = 2, DFLO_SELF_IPC = 3 } dflo_self_transport;
int dflo_hip_multi_create_self(const dflo_mesh_t *mesh, const dflo_params_t *params, int device_id, int n_virtual, int partitioner,
                               int transport, dflo_hip_multi_handle *out);
int dflo_hip_multi_n_parts(dflo_hip_multi_handle m);            /* parts of the partition */
int dflo_hip_multi_n_local(dflo_hip_multi_handle m);            /* parts (engines) this process owns */
dflo_hip_handle dflo_hip_multi_engine(dflo_hip_multi_handle m, int i); /* i-th local engine (timing, inspection) */
int dflo_hip_multi_part_cells(dflo_hip_multi_handle m, int i, int32_t *n_owned, int32_t *n_ghost, const int64_t **global_ids);
int64_t dflo_hip_multi_n_dofs(dflo_hip_multi_handle m);         /* of the undivided mesh */
int64_t dflo_hip_multi_n_owned_dofs(dflo_hip_multi_handle m);   /* owned by this process */
int32_t dflo_hip_multi_n_rk(dflo_hip_multi_handle m);
/* The same for one local part in ITS numbering (owned cells first, then its ghost cells; dflo_hip_multi_part_mesh
 * gives the part's mesh, owned by the handle): a rank of a large run evaluates the initial data on its own cells only,
 * as VectorTools::interpolate does on the locally owned range (src_mpi/ic.cc). */
const dflo_mesh_t *dflo_hip_multi_part_mesh(dflo_hip_multi_handle m, int i);
int dflo_hip_multi_set_part_solution(dflo_hip_multi_handle m, int i, const double *u_part);
int dflo_hip_multi_synchronize(dflo_hip_multi_handle m);
/* Reporting (bench.py's N > 1 line): the average time, in microseconds, that the comm stream of the local parts spent in an
 * exchange of halo records (every fifth exchange is bracketed by events: one process per GPU -- the grouped send / receive,
 * the rendezvous with the peers included; one process -- the wait for the peers' records), and what the transport is: the
 * number of ranks and this process's rank AS THE RCCL COMMUNICATOR REPORTS THEM (ncclCommCount / ncclCommUserRank; the
 * partition's numbers for the other transports, rank -1 in one process) and a description of the transport in use. */
int dflo_hip_multi_exchange_timing(dflo_hip_multi_handle m, int enable, double *avg_us, int64_t *n);
int dflo_hip_multi_comm_info(dflo_hip_multi_handle m, int32_t *comm_count, int32_t *comm_rank, char *transport, int32_t transport_len);

#ifdef __cplusplus
}
#endif
#endif /* DFLO_HIP_TRANSPORT_H */
