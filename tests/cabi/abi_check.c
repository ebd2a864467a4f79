/* Plain-C consumer of include/ufr.h: the header must compile as C, every declared entry point must link against
 * libufr.so, and the argument checks must answer without a GPU.  Built and run by tests/test_abi_and_layout.py. */
#include <stdio.h>
#include <string.h>

#include "ufr.h"

int main(void) {
  /* link-time presence of every entry point */
  /* every function include/ufr.h declares: the list is GENERATED from the header by tests/test_abi_and_layout.py
   * (abi_syms.inc in the build directory), so a new entry point cannot be forgotten here */
  const void* syms[] = {
#include "abi_syms.inc"
  };
  unsigned i, n = sizeof(syms) / sizeof(syms[0]);
  for (i = 0; i < n; ++i)
    if (!syms[i]) return 10;
  if (ufr_version() != UFR_ABI_VERSION) return 11;
  if (ufr_get_matrix_precision() != UFR_PRECISION_FP32 || ufr_set_matrix_precision(7) == 0) return 15;
  /* argument validation: negative status + message, no device needed */
  if (ufr_sample_fixed(0, 0, 0, 0, 4, 64, 0) >= 0) return 12;
  if (!strstr(ufr_last_error(), "ufr_sample_fixed")) return 13;
  if (ufr_tsdf_integrate(0, 0, 0, 0, 0, 1.f, 1.f, 0, 0, 0, 0, 4, 4, 1.f, 0, 0) >= 0) return 14;
  if (ufr_deform_conv2d(0, 0, 0, 0, 0, 0, 1, 32, 32, 8, 8, 0, 0, 0) >= 0) return 15;
  if (ufr_render_workspace_bytes(4096, 64, 64, 3) == 0) return 16;
  if (ufr_composite_bwd(0, 0, 0, 0, 0, 4, 64, 0, 0, 0, 0, 0, 0, 0, 0, 0) >= 0) return 18;
  if (ufr_render_loss(0, 0, 0, 0, 0, 0, 0, 2, 1, 4, 1.f, 1.f, 0, 0, 0, 0, 0, 0) >= 0) return 31;
  if (ufr_aggregate_bwd(0, 0, 0, 0, 0, 0, 0, 4, 64, 3, 0, 0, 0, 0, UFR_PRECISION_DEFAULT, 0) >= 0) return 19;
  /* an unknown precision is an argument error, not a silent default */
  if (ufr_view_transform((const void*)1, (const float*)1, (const float*)1, (const float*)1, 4, 3, (float*)1, (float*)1, 7, 0) != UFR_ERR_ARG) return 21;
  if (ufr_aggregate_bwd_workspace_bytes(1024, 128, 3) == 0) return 20;
  if (ufr_correlate_workspace_bytes(32, 128, 160, 2) == 0) return 17;
  printf("abi ok: %u entry points, version %d, %zu packed weight bytes\n", n, ufr_version(), ufr_packed_weights_bytes());
  return 0;
}
