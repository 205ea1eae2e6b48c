/*
 * include/ltrace_probes.h -- diagnostic microbenchmarks (VALU issue costs, the bare RK4 step, pieces
 * of the right-hand side).  NOT part of the product: they are compiled only with -DLT_PROBES into
 * light-path-tracer_amd/lib/libltrace_probes.so (`python __graft_entry__.py --probes`), which also
 * exports everything ltrace.h declares.  tools/probes.py binds them.
 */
#ifndef LTRACE_PROBES_H
#define LTRACE_PROBES_H
#include "ltrace.h"
#ifdef __cplusplus
extern "C" {
#endif

/* FP32 VALU issue-rate microbenchmark used to calibrate the roofline: runs `iters` dependent-chain
 * FMA blocks per lane; mode 0 = v_fma_f32, 1 = v_pk_fma_f32.  Returns achieved TFLOP/s in *tflops. */
int lt_valu_peak_probe(int mode, int iters, double *tflops);

/* VALU issue-cost microbenchmark (diagnostic; DESIGN.md "issue-rate roofline"): instruction class
 * `index` in [0, lt_valu_issue_probe_count()), `waves_per_simd` resident waves per SIMD (1..8).
 * constant_data != 0 runs it on all-equal operands (no datapath toggling: highest clock).
 * Returns ns per wave-instruction per SIMD and the shader clock the chip held during the loop
 * (s_memtime / s_memrealtime); name_out receives the instruction mnemonic. */
int lt_valu_issue_probe(int index, int waves_per_simd, int iters, int constant_data, char *name_out,
                        int name_len, double *ns_per_instr, double *clock_mhz);
int lt_valu_issue_probe_count(void);
/* The Kerr RK4 step alone (no events, no divergence), `iters` times per lane, `waves_per_simd`
 * resident waves per SIMD: shader cycles one SIMD spends per wave-step, and the clock held. */
/* Pieces of the right-hand side (0 sincos, 1 the rest, 2 the rest without the reciprocal): SIMD
 * cycles per evaluation per wave. */
int lt_piece_probe(int piece, int waves_per_simd, int iters, double *cycles_per_eval, double *clock_mhz);
int lt_rk4_step_probe(int precision, int waves_per_simd, int iters, double *cycles_per_wave_step,
                      double *clock_mhz);

#ifdef __cplusplus
}
#endif
#endif /* LTRACE_PROBES_H */
