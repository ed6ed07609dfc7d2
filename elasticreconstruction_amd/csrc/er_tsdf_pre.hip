// er_tsdf_pre.hip -- translation unit 1 of er_tsdf.hip: the pre-pass kernels k_reproject_scatter and k_prepare, compiled with their own
// flags (Makefile: FLAGS_er_tsdf_pre.hip).  See the note on ER_TSDF_TU at the top of er_tsdf.hip.
#define ER_TSDF_TU 1
#include "er_tsdf.hip"
