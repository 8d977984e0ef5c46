/* clarabel_hip_testing.h -- test hooks of libclarabel_hip.so.  NOT part of the drop-in boundary (clarabel_hip.h):
 * these entry points exist only in libraries built with -DCHIP_TESTING (the in-tree default of csrc/Makefile, which
 * the test-suite needs; `make TESTING=0` leaves them out). */
#ifndef CLARABEL_HIP_TESTING_H
#define CLARABEL_HIP_TESTING_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif
/* launches a kernel of `blocks` x `threads` (+ lds_bytes of LDS per workgroup) that only spins for `usec`
 * microseconds, on a private stream of `device` -- the persistent launches (k_bundle_ir, k_gstep_*) must survive a
 * co-resident kernel (the RCCL ring of the sharded path).  blocks = 0: waits for the spinners launched so far.
 * The calling thread's current device is left as it was. */
int32_t chip_debug_spin(int32_t device, int32_t blocks, int32_t threads, int32_t lds_bytes, double usec);
/* sets (value != NULL) or clears one CHIP_* diagnostic switch by its environment name and re-parses the switch
 * table (csrc/switches.hpp) -- for flipping a switch on handles that already exist; switches given in the
 * environment are picked up whenever a handle is created.  Returns CHIP_ERR_ARG for an unknown name. */
int32_t chip_debug_set_switch(const char *name, const char *value_or_null);
/* one structural figure of a KKT handle (opaque chip_kkt *, clarabel_hip.h) by name, for tests that must know which
 * mechanism a handle ended up with: "dense_blocks" / "dense_block_rows" (dense diagonal blocks of the top that the
 * residual multiplies from K's values directly), "nnzS" (entries of the full-row copy of the top rows), "psd_hs_row_blocks" (> 0: the PSD cones write Hs row by row
 * of the value store, k_psd_write_hs_rows),
 * "assembled_levels" (unit levels whose ancestor updates can be assembled per target column), "assembled_targets".
 * Returns CHIP_ERR_ARG for an unknown name. */
int32_t chip_debug_counter(const void *kkt_handle, const char *name, double *out);
/* a spinner of `blocks` x `threads` for `usec` microseconds on the stream of a communicator (opaque chip_comm *), behind
 * the collective enqueued last; the communicator's completion event moves behind it.  On one GPU this stands in for
 * the time RCCL's ring kernel holds CUs when several ranks exchange (bench.py --coresident). */
int32_t chip_comm_debug_spin(void *comm, int32_t blocks, int32_t threads, double usec);
#ifdef __cplusplus
}
#endif
#endif
