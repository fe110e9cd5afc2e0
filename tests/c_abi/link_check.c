/* Plain-C consumer of the drop-in boundary: include/temx.h must compile as C (no C++ or torch
 * types) and every declared entry point must resolve against libtemx.so.  No GPU needed. */
#include <stdio.h>
#include "temx.h"

int main(void) {
  const void* syms[] = {
      (const void*)temx_version, (const void*)temx_last_error, (const void*)temx_device_count,
      (const void*)temx_plan_create, (const void*)temx_plan_finalize, (const void*)temx_plan_refine, (const void*)temx_plan_set_weights,
      (const void*)temx_plan_destroy, (const void*)temx_plan_is_paired, (const void*)temx_plan_sweep_mode, (const void*)temx_plan_one_pass, (const void*)temx_plan_single_sweep,
      (const void*)temx_get_matrix, (const void*)temx_project, (const void*)temx_zonal_mean,
      (const void*)temx_zonal_mean_from_sums, (const void*)temx_plan_set_tem, (const void*)temx_tem_stage1,
      (const void*)temx_tem_stage2, (const void*)temx_tem_stage2_from_sums, (const void*)temx_tem_stage3, (const void*)temx_tem_run,
      (const void*)temx_tem_eddy, (const void*)temx_tem_eddy_rows, (const void*)temx_tracer_stage1, (const void*)temx_tracer_stage2,
      (const void*)temx_tracer_stage3, (const void*)temx_tracer_stage1_sums, (const void*)temx_tracer_stage2_from_sums,
      (const void*)temx_tracer_run, (const void*)temx_tem_tracer_stage1, (const void*)temx_tem_tracer_run, (const void*)temx_tracer_eddy,
      (const void*)temx_status, (const void*)temx_synth_fields, (const void*)temx_mfma_f64_peak,
      (const void*)temx_kernel_timing, (const void*)temx_kernel_timing_read,
      (const void*)temx_plan_configure, (const void*)temx_plan_option, (const void*)temx_plan_set_os_matrices,
      (const void*)temx_tem_os_prepass, (const void*)temx_tem_os_sweep, (const void*)temx_tem_os_tail,
      (const void*)temx_tracers_os_prepass, (const void*)temx_tracers_os_sweep, (const void*)temx_tracers_os_tail, (const void*)temx_tracers_run,
      (const void*)temx_tem_tail_from_sums, (const void*)temx_time_slices, (const void*)temx_selftest_exception};
  unsigned n = (unsigned)(sizeof(syms) / sizeof(syms[0])), i, ok = 0;
  for (i = 0; i < n; ++i) ok += syms[i] != 0;
  /* argument checking happens before any device call: a null plan is an error, not a crash */
  int rc = temx_plan_set_tem(0, 2, 1, 0, 101325.0);
  printf("temx_version=%d symbols=%u/%u null_plan_rc=%d err=\"%s\"\n", temx_version(), ok, n, rc, temx_last_error());
  return (ok == n && temx_version() > 0 && rc == TEMX_EINVAL) ? 0 : 1;
}
