/* placeholder until the 3-D maze restatement lands */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
void orc_mc3d_stats(const uint8_t *grid, int Z, int Y, int X, int32_t *stats, int16_t *path_xyz, int32_t *path_len) {
  (void)grid; (void)Z; (void)Y; (void)X; (void)stats; (void)path_xyz; (void)path_len;
  fprintf(stderr, "orc_mc3d_stats: not implemented\n");
  abort();
}
