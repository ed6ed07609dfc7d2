/* The C ABI from plain C (C99): version, device count, and the error path of a constructor when there is no GPU.
 *   gcc -std=c99 -Iinclude examples/abi_probe.c -Lelasticreconstruction_amd -ler_hip -Wl,-rpath,$PWD/elasticreconstruction_amd -o abi_probe */
#include <stdio.h>

#include "er_hip.h"

int main(void) {
  er_tsdf_t vol = NULL;
  int n = er_device_count();
  printf("abi %d, %d HIP device(s)\n", er_abi_version(), n);
  if (er_tsdf_create(640, 480, NULL, 16, 0, &vol) != 0) {
    printf("er_tsdf_create: %s\n", er_last_error());        /* no CPU fallback: this is what a GPU-less host sees */
    return n == 0 ? 0 : 1;
  }
  {
    int units = -1;
    er_tsdf_unit_count(vol, &units);
    printf("empty volume: %d units\n", units);
    er_tsdf_destroy(vol);
  }
  return 0;
}
