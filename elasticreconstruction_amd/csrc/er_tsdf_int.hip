// er_tsdf_int.hip -- translation unit 2 of er_tsdf.hip: the voxel pass k_integrate, compiled with its own flags (Makefile:
// FLAGS_er_tsdf_int.hip).  See the note on ER_TSDF_TU at the top of er_tsdf.hip.
#define ER_TSDF_TU 2
#include "er_tsdf.hip"
