/* er_hip.h -- C ABI of liber_hip.so, the MI355X (gfx950) implementation of the two data-parallel
 * stages of qianyizh/ElasticReconstruction:
 *
 *   path A  TSDF depth integration with control-grid warp   (reference: Integrate/)
 *   path B  pairwise ICP refinement + correspondences       (reference: BuildCorrespondence/)
 * and, after those two, the first rows of SURVEY.md 8f:
 *   RansacCurvature::getFitness / getInformation             (reference: GlobalRegistration/RansacCurvature.h)
 *   COptApp's per-point state, Hessian assembly and solve    (reference: FragmentOptimizer/OptApp.cpp, PointCloud.h)
 *
 * The reference has no FFI; its boundary is "executable + argv + files" (SURVEY.md 8b).  These
 * entry points are the thin C ABI the new host programs (and any cgo/JNI/ctypes caller) bind.
 * Each one cites the reference method it replaces.  Conventions:
 *   - plain C types only; 4x4 matrices are ROW-MAJOR arrays of 16 (double unless stated);
 *   - every function returns 0 on success, non-zero on failure; er_last_error() (thread-local)
 *     explains the last failure; nothing throws across the ABI;
 *   - handles are opaque and owned by the caller; calls on ONE handle must be serialised by the
 *     caller, different handles may be driven from different host threads (this mirrors the
 *     reference's "one OpenMP thread per pair" model, CorresApp.cpp:121,220);
 *   - "host" pointers are ordinary host memory, "dev" pointers are HIP device memory on the
 *     handle's device; there is NO CPU fallback: without a usable HIP device every constructor
 *     fails with an error.
 */
#ifndef ER_HIP_H_
#define ER_HIP_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ER_UNIT_RES 64                 /* TSDFVolume.cpp:57: new TSDFVolumeUnit( 64, ... ) */
#define ER_UNIT_VOX (64 * 64 * 64)
#define ER_MAX_BATCH 64                /* frames fused per launch (one bit each in a 64-bit mask) */

/* ------------------------------------------------------------------ misc ---- */
const char* er_last_error(void);
int er_device_count(void);                       /* number of visible HIP devices (0 if none) */
int er_abi_version(void);
/* OPT-IN, call before the process's first HIP call: asks the HIP runtime for n hardware queues (GPU_MAX_HW_QUEUES, default 4;
 * n <= 0 means 8) without overriding a value the user exported; returns the value in force.  A TSDF volume overlaps four
 * streams and runs ~15 % slower when two of them share a queue.  The library never sets this on its own (it would change the
 * queue set-up of every GPU user of the process): the drop-in programs and bench.py call it first thing in main(). */
int er_request_hw_queues(int n);

/* Page-locked host memory.  Every "host" pointer of this ABI may be ordinary (pageable) memory; result buffers
 * that come from er_host_alloc are written by asynchronous device-to-host copies without a staging pass. */
void* er_host_alloc(size_t bytes);               /* NULL on failure (see er_last_error) */
int er_host_free(void* p);
int er_host_copy_h2d(void* dev_dst, const void* host_src, size_t bytes);   /* blocking host -> device copy (current device) */
/* Device memory for hosts that do not speak HIP themselves: the list buffers of er_find_correspondence_batch / er_registration_batch
 * may live in HBM (the lists then never cross PCIe) and er_fopt_set_correspondences_dev consumes them there; er_device_copy_d2h fetches
 * one when a corres_<i>_<j>.txt is to be written after all. */
void* er_device_alloc(size_t bytes, int device);  /* NULL on failure */
int er_device_free(void* p);
int er_device_copy_d2h(void* host_dst, const void* dev_src, size_t bytes);   /* blocking device -> host copy */

/* ------------------------------------------------------------ path A: TSDF ---- */
typedef struct er_tsdf_s* er_tsdf_t;

/* Control-grid warp inputs for a batch of n frames (CIntegrateApp::Reproject, IntegrateApp.cpp:228-269).
 * All pointers are HOST memory. */
typedef struct er_warp {
  const float* ctr;         /* num_grids * (resolution+1)^3 * 3 floats; vertex i + j*(res+1) + k*(res+1)^2
                               (ControlGrid.h:41-43; file order of ControlGrid::Load, ControlGrid.cpp:15-34) */
  int num_grids;            /* --num        (IntegrateApp.h:70 ctr_num_) */
  int resolution;           /* --resolution (IntegrateApp.h:68) */
  float length;             /* --length, stored as float by ControlGrid::Load (ControlGrid.cpp:17-18) */
  const int* grid_index;    /* n ints: chunk = (frame_id-1)/interval of each frame (IntegrateApp.cpp:242) */
  const double* seg;        /* n * 16: seg_traj_[frame_id-1] (IntegrateApp.cpp:251) */
  const double* madj;       /* n * 16: traj[f-1]^-1 * traj[0] * seg[0]^-1 (IntegrateApp.cpp:243) */
} er_warp;

/* TSDFVolume::TSDFVolume( cols, rows ) + CameraParam (TSDFVolume.cpp:7-13, TSDFVolumeUnit.h:65-70).
 * cam6 = fx fy cx cy ICP_trunc integration_trunc (NULL = reference defaults 525 525 319.5 239.5 2.5 2.5).
 * max_units = capacity of the 64^3 volume-unit pool in HBM (2 MiB each, zero-filled up front). */
int er_tsdf_create(int cols, int rows, const float cam6[6], int max_units, int device, er_tsdf_t* out);
int er_tsdf_destroy(er_tsdf_t h);

/* Run the handle's voxel pass and every other call on an existing hipStream_t (e.g. torch's current stream).
 * NULL = the handle's own stream.  (The pre-passes always use two internal auxiliary streams.) */
int er_tsdf_set_stream(er_tsdf_t h, void* hip_stream);
int er_tsdf_synchronize(er_tsdf_t h);

/* TSDFVolume::ScaleDepth (TSDFVolume.cpp:19-36): depth uint16[rows*cols] -> scaled float[rows*cols]. */
int er_tsdf_scale_depth(er_tsdf_t h, const uint16_t* depth_host, float* scaled_host);

/* CIntegrateApp::Reproject pixel loop (IntegrateApp.cpp:236-268) for ONE frame, in place on host memory. */
int er_tsdf_reproject(er_tsdf_t h, uint16_t* depth_inout_host, const float* ctr_host, int resolution, float length,
                      const double seg[16], const double madj[16]);

/* One frame of CIntegrateApp::Execute's numeric tail (IntegrateApp.cpp:224-225):
 * ScaleDepth + TSDFVolume::Integrate (TSDFVolume.cpp:38-67 -> IntegrateVolumeUnit :69-102). */
int er_tsdf_integrate(er_tsdf_t h, const uint16_t* depth_host, const double T[16]);

/* n frames in order: [Reproject] + ScaleDepth + Integrate for each (IntegrateApp.cpp:217-225).
 * Results are identical to n sequential er_tsdf_integrate calls; internally frames are fused
 * ER_MAX_BATCH at a time so each voxel is read and written once per batch.
 * depth: n*rows*cols uint16, host memory if depth_on_device == 0 else device memory (left untouched).
 * Device depth must be COMPLETE when the call is made (synchronise the stream that produced it): the
 * per-pixel pre-passes of the next two batches run on the handle's two auxiliary streams so that they overlap each other
 * and the voxel pass of the previous batch, and those streams do not wait for the caller's (er_tsdf_wait_event).
 * T: n*16 host doubles (traj_[frame_id-1]).  warp may be NULL (rigid, --ref_traj). */
int er_tsdf_integrate_frames(er_tsdf_t h, int n, const uint16_t* depth, int depth_on_device, const double* T,
                             const er_warp* warp);

/* Device depth produced asynchronously: the pre-passes of the NEXT er_tsdf_integrate_frames calls (both internal pre-pass
 * streams) wait for this hipEvent_t (recorded by the caller after the producer of the depth buffer) instead of requiring a
 * synchronised stream.  (Host depth needs no event: it is read by the copy inside the call.) */
int er_tsdf_wait_event(er_tsdf_t h, void* hip_event);

/* Empties the volume (data_.clear()): every unit is released and zero-filled, the hash map is emptied, flags are cleared.
 * Buffers, streams and the unit-shard setting are kept. */
int er_tsdf_reset(er_tsdf_t h);

/* Sticky error flags + the number of depth pixels skipped because their unit index left [0,512) (> +-96 m, where the
 * reference's hash_key would alias another unit).  A POLL: it does not wait for queued batches, so a program can call it
 * after every er_tsdf_integrate_frames and fail fast instead of learning about an exhausted pool in SaveWorld. */
#define ER_STATUS_POOL_EXHAUSTED 1
#define ER_STATUS_TABLE_FULL 2
int er_tsdf_status(er_tsdf_t h, int* flags, long* out_of_range_pixels);

/* data_ map access (TSDFVolume.h:27) as used by SaveWorld. */
int er_tsdf_unit_count(er_tsdf_t h, int* count);
int er_tsdf_unit_keys(er_tsdf_t h, int* keys_host);                  /* ascending hash_key order */
int er_tsdf_read_unit(er_tsdf_t h, int key, float* sdf_host, float* weight_host);  /* 64^3 floats each */

/* Sum of weight_ over the volume = number of voxel updates so far (TSDFVolume.cpp:90,94). */
int er_tsdf_sum_weight(er_tsdf_t h, double* sum);

/* TSDFVolume::SaveWorld's voxel filter (TSDFVolume.cpp:104-132) on device.  Writes (x,y,z,intensity)
 * float4 per point, units in ascending key order, voxels in i,j,k order.  out_host may be NULL to query
 * the count; capacity is in points. */
int er_tsdf_extract_world(er_tsdf_t h, float* out_host, long capacity, long* count);

/* Zero-crossing points of the resident volume (the consumer of world.pcd, kinfu's mesh/point output, does this
 * on the host after reading the file back): for every observed voxel and its +x / +y / +z neighbour (also across unit
 * borders), both with weight != 0 and sdf of strictly opposite sign, the point where the surface crosses the lattice edge,
 * p = voxel position + F / (F - Fn) * voxel size along that axis (metres, float32; position = (float)(global index * 3/512)).
 * float4 per point = x y z axis(0,1,2); units in ascending key order, voxels in i,j,k order, axes x,y,z.
 * out_host may be NULL to query the count; capacity is in points. */
int er_tsdf_extract_surface(er_tsdf_t h, float* out_host, long capacity, long* count);

/* Marching cubes on the resident volume (the triangle connectivity the pipeline's next step -- kinfu's mesh output,
 * outside the reference repository -- builds from world.pcd).  A cell of 2 x 2 x 2 voxels yields triangles only if all eight are
 * observed (weight != 0); a corner is inside iff sdf < 0; vertices are the linear zero crossings on the lattice edges, in metres.
 * tri_host: 9 floats per triangle (three vertices x, y, z; the normal of the winding points towards positive sdf = free space);
 * order: units by ascending key, cells in i, j, k order, triangles in table order.  Vertices shared by neighbouring cells are
 * bit-identical, so the soup can be welded by exact comparison.  tri_host == NULL: only counts.  er_mc_table copies the 256 x 16
 * case table (three edge ids per triangle, 255-terminated; conventions in csrc/er_mc_table.h) -- host only, no GPU needed. */
int er_tsdf_extract_mesh(er_tsdf_t h, float* tri_host, long capacity_triangles, long* n_triangles);
int er_mc_table(unsigned char out[256 * 16]);

/* Multi-GPU frame split: for the given key list write sum-ready planes into dev_buf
 * (n_keys * 2 * 64^3 floats: [key][0] = sdf*weight, [key][1] = weight; zeros for keys absent here), and
 * after an external all-reduce(sum) read them back as weight = W, sdf = SW / W. */
int er_tsdf_export_weighted(er_tsdf_t h, const int* keys_host, int n_keys, float* dev_buf);
int er_tsdf_import_weighted(er_tsdf_t h, const int* keys_host, int n_keys, const float* dev_buf);
/* The same layout with [key][0] = sdf itself: a unit BIT FOR BIT -- how a unit that only one GPU touched travels (TSDFVolume.cpp:93-94 is a
 * sum only where frames of two GPUs met; sdf * w / w would round the others for nothing).  import_raw creates the unit if it is absent and
 * overwrites it otherwise. */
int er_tsdf_export_raw(er_tsdf_t h, const int* keys_host, int n_keys, float* dev_buf);
int er_tsdf_import_raw(er_tsdf_t h, const int* keys_host, int n_keys, const float* dev_buf);
/* Round 6, BAND RECORDS: a unit as its observed voxels only (weight_ != 0: the truncation band and the free space in front of it, about a quarter of a
 * touched unit), the sdf_ only where it is not exactly 1 (free space: four observed voxels out of five) and the weight_ as 16 bits when every weight of
 * the unit is a frame count below 65536.  A never-updated voxel is (+0, 0) (TSDFVolumeUnit.cpp:4-21), so a record restores a unit bit for bit.
 * Records are DEVICE memory, 8-byte aligned, a whole number of 32-bit words (layout: csrc/er_tsdf.hip, "band records").  band_sizes: the record size in
 * words of each unit this GPU holds (a key it does not hold is an error).  export_band: the records of the given units back to back into dev_block
 * (sizes as band_sizes returned them; a volume that changed in between is an error).  merge_band: the OWNER's step of the frame-split merge -- unit
 * keys[u] becomes the sum of its own voxels and nsrc[u] <= 16 records (recs[16 u + k], the other GPUs' in rank order; self_pos[u] of them come before
 * its own voxels in that order): SW = sum fl(sdf_r w_r), W = sum w_r, sdf = SW / W, TSDFVolume.cpp:93-94 as a sum in an order fixed by the arguments.
 * import_band: create / overwrite units from records.  drop_units: this GPU hands the units over -- they are zeroed and disappear from
 * er_tsdf_unit_count / unit_keys / read_unit / the extractions until the next integrated frame or import touches them.  All of them run on the handle's
 * stream and return when the device work is done. */
int er_tsdf_band_sizes(er_tsdf_t h, const int* keys_host, int n_keys, int* words_host);
int er_tsdf_export_band(er_tsdf_t h, const int* keys_host, const int* words_host, int n_keys, void* dev_block);
int er_tsdf_merge_band(er_tsdf_t h, const int* keys_host, int n_keys, const int* nsrc, const int* self_pos, const void* const* recs);
int er_tsdf_import_band(er_tsdf_t h, const int* keys_host, int n_keys, const void* const* recs);
int er_tsdf_drop_units(er_tsdf_t h, const int* keys_host, int n_keys);

/* Multi-GPU, the bit-exact alternative: shard the volume BY UNIT.  Every GPU is fed ALL frames and runs
 * their pre-pass, but only allocates / integrates / reports the units with er_unit_owner(key, world) == rank.  Units are disjoint
 * (TSDFVolume.cpp:45-63) and each still sees every frame in order, so the union of the GPUs' volumes equals the single-GPU
 * volume bit for bit and no collective touches the volume.  Must be set before the first frame. */
int er_tsdf_set_unit_shard(er_tsdf_t h, int rank, int world);
int er_unit_owner(int key, int world);          /* (xi + yi + zi) mod world: diagonal stripes of the unit lattice */

/* Multi-GPU frame split behind the C ABI (the Python path over torch.distributed is parallel.py).  One
 * communicator per GPU over RCCL (xGMI inside a node; librccl is loaded on first use):
 *   er_comm_unique_id + er_comm_create   one PROCESS per GPU: rank 0 draws the 128-byte id and hands it to the other ranks
 *                                        out of band (a file, a socket, torch.distributed), then every rank creates its
 *                                        communicator (ncclCommInitRank; collective: all ranks must call it);
 *   er_comm_create_local                 ONE process, n GPUs, one host thread per GPU afterwards (ncclCommInitAll).
 * er_tsdf_allreduce merges the private volumes of the ranks after each integrated its own contiguous frame block
 * (er_frame_block).  Every rank calls it once.  Since round 6 it is the OWNER MERGE (csrc/er_merge_protocol.h: merge_protocol_owner), a
 * reduce-scatter by volume unit:
 *   agree on the key count; all-gather the touched unit keys with their observed-voxel counts -- every rank then knows who touched what and how
 *   much; every unit of the union gets an OWNER, the toucher that observed most of it; every other toucher sends the owner a band record of
 *   its voxels (above) in ONE grouped ncclSend / ncclRecv step -- each (sender, owner) pair is an xGMI link of its own --; the owner adds the
 *   records to its own voxels IN RANK ORDER in one kernel: weights exact, sdf within 1e-5 of the sequential running mean of TSDFVolume.cpp:93-94
 *   (the float32 summation order differs from the frame order), and bit-reproducible: the order is a function of the key sets, not of the wire;
 *   the non-owners drop their copies.
 * root == ER_MERGE_DISTRIBUTED stops there: the merged volume stays distributed, every unit complete on exactly one rank (SaveWorld is per unit,
 * TSDFVolume.cpp:104-132: bin/Integrate --gpus N concatenates the ranks' extractions by key).  root >= 0: the owners then send their finished
 * units, as band records, to `root` in a second grouped step; root == ER_MERGE_ALL (-1): to every other rank.  Units only ONE rank touched are
 * that rank's: they stay where they are (distributed) or travel bit for bit (records restore a unit exactly).
 * ER_MERGE_IMPL=ring in the environment of EVERY rank selects round 5's protocol instead (ONE ncclReduce / ncclAllReduce over whole
 * [sdf*weight | weight] planes of the units two or more ranks touched, summed in RCCL's order; raw point-to-point for the others; root >= 0 or -1 only).
 * union_units (nullable) receives the size of the key union; er_comm_merge_stats what the last merge of this communicator moved: stats[0] union,
 * [1] multi-toucher units, [2] single-toucher units, [3] units this rank sent (owner merge: handed over), [4] units this rank received (owner merge:
 * summed here), [5] bytes handed to a reduction collective (owner merge: 0), [6] bytes this rank sent, [7] bytes received.  er_comm_merge_stats_owner:
 * [0] protocol of the last merge (0 ring, 1 owner), [1] union, [2] multi-toucher, [3] single-toucher, [4] units this rank owns afterwards, [5] of
 * which it summed, [6] units it handed over, [7] / [8] bytes sent / received in the step to the owners, [9] / [10] in the step to the root /
 * everybody, [11] bytes round 5's ring reduction would have been handed for the same key sets (2 MiB per multi-toucher unit). */
typedef struct er_comm_s* er_comm_t;
#define ER_COMM_ID_BYTES 128
int er_comm_unique_id(unsigned char id[ER_COMM_ID_BYTES]);
int er_comm_create(const unsigned char id[ER_COMM_ID_BYTES], int rank, int world, int device, er_comm_t* out);
int er_comm_create_local(int n, const int* devices, er_comm_t* out /* n handles */);
/* n <= 16 ranks = n host threads of this process, ALL on one device, no RCCL: the same merge protocol over the same device volumes and
 * export / import kernels, the sum reduction as a kernel that adds the ranks' plane buffers in rank order, the point-to-point step as
 * device-to-device copies.  For boxes with one GPU (RCCL refuses two ranks on one device): `Integrate --gpus N --same_device`, tests. */
int er_comm_create_loopback(int n, int device, er_comm_t* out /* n handles */);
int er_comm_destroy(er_comm_t c);
int er_comm_rank(er_comm_t c);
int er_comm_world(er_comm_t c);
#define ER_MERGE_ALL (-1)
#define ER_MERGE_DISTRIBUTED (-2)
int er_tsdf_allreduce(er_tsdf_t h, er_comm_t c, int root, int* union_units);
int er_comm_merge_stats(er_comm_t c, long long stats[8]);
int er_comm_merge_stats_owner(er_comm_t c, long long stats[12]);
/* Contiguous frame block [lo, hi) of `rank` (IntegrateApp.cpp:190-226 is the loop being split). */
void er_frame_block(int n_frames, int rank, int world, int* lo, int* hi);

/* Kernel timing: HIP events on the handle's stream around every `enable`-th IntegrateVolumeUnit launch (1 = every launch, 0 = off;
 * each timed launch puts two more packets on the stream, about 2 % of the frame rate when every launch is timed).  get_profile
 * returns the sum and the number of the TIMED launches, all frames and all unit visits since set_profiling. */
int er_tsdf_set_profiling(er_tsdf_t h, int enable);
int er_tsdf_get_profile(er_tsdf_t h, double* integrate_ms_total, long* integrate_launches, long* frames,
                        long* unit_visits);

/* ------------------------------------------------------------- path B: ICP ---- */
typedef struct er_cloud_s* er_cloud_t;

/* pointclouds_[i] after the NaN-normal filter (CorresApp.cpp:94-98): n points, xyz and normals as
 * separate float[3*n] host arrays (AoS xyz xyz ...).  grid_cell = edge of the uniform search grid
 * built over this cloud when it is used as a TARGET; must be >= every max distance queried later
 * (use reg_dist_, CorresApp.cpp:16).  n < 2^27 (134 217 728) points: the search kernels address a cloud with 32-bit byte offsets; larger
 * clouds are refused. */
int er_cloud_create(const float* xyz_host, const float* normal_host, int n, float grid_cell, int device,
                    er_cloud_t* out);
/* The clouds of a LIST of fragments in one call (LoadData's loop, CorresApp.cpp:82-110): xyz_host[i] / normal_host[i] hold counts[i] points.
 * All uploads are queued up front (coordinates and normals on two copy streams); chunks of up to 8 clouds share their device allocations
 * and ONE set of grid launches, built while the later chunks are still uploading; the host waits once per chunk.  Page-locked input
 * arrays (er_host_alloc) make the uploads asynchronous: the list is then bound by PCIe (6 x 4 bytes per point), not by the host.
 * All or nothing: on failure no cloud is left behind.  out[i] as er_cloud_create; the clouds may be destroyed in any order (a chunk's
 * allocations go with its last cloud). */
int er_cloud_create_batch(int n_clouds, const float* const* xyz_host, const float* const* normal_host, const int* counts, float grid_cell,
                          int device, er_cloud_t* out);
int er_cloud_destroy(er_cloud_t c);
int er_cloud_size(er_cloud_t c);

/* Registration pre-check (CorresApp.cpp:249-264): cnt = #{k : NN sqdist(T*src[k], tgt) < max_dist^2}. */
int er_icp_count_inliers(er_cloud_t src, er_cloud_t tgt, const double T[16], double max_dist, int* count);

/* pcl::IterativeClosestPoint::align with TransformationEstimationPointToPlaneLLS as configured at
 * CorresApp.cpp:295-306 (source = src, target = tgt).  guess/out are ROW-MAJOR FLOAT 4x4.
 * stop_rule: 0 = PCL 1.7 DefaultConvergenceCriteria, 1 = PCL <= 1.6 rule.
 * fitness (may be NULL) = mean squared NN distance within max_dist after the final transform. */
int er_icp_align(er_cloud_t src, er_cloud_t tgt, const float guess[16], double max_dist, int max_iter,
                 double transformation_epsilon, int stop_rule, float out[16], int* iterations, int* converged,
                 double* fitness);

/* FindCorrespondence (CorresApp.cpp:144-161, 186-208): pairs (tgt_index, src_index) ascending in
 * src_index, for NN sqdist < dist^2 and normal dot > normal_cos; info36 (nullable) = row-major 6x6
 * information matrix over the untransformed source points. pairs_host holds 2*capacity ints -- pageable host memory,
 * page-locked host memory (er_host_alloc: no staging pass) or, since round 5, DEVICE memory of the clouds' GPU (er_device_alloc:
 * the list stays in HBM, a device-to-device copy of exactly the list; the same holds for every pairs_host of the *_batch forms). */
int er_find_correspondence(er_cloud_t src, er_cloud_t tgt, const double T[16], double dist, double normal_cos,
                           int* pairs_host, int capacity, int* n_pairs, double* info36);

/* The reference runs its two loops over the pair list with "#pragma omp parallel for" (CorresApp.cpp:121,220).
 * The *_batch forms take the whole list of one loop: n pairs (src[i], tgt[i]) on ONE device, per-pair inputs and
 * outputs as arrays (T: n*16 doubles, guess/out: n*16 floats, info36: n*36 doubles, ...).  Results are those of n single
 * calls, bit for bit: integers (counts, iteration counts, correspondence lists) and -- since round 6 -- every float of every transform.  The 29
 * float64 sums of an ICP iteration are added per wave of 64 consecutive points (a function of the pair alone) and from there on as 64-bit fixed-point
 * integers whose power-of-two scales come from certain bounds of the data, so the totals do not depend on how the library groups the work: not on the
 * list a pair is part of, its position, ER_ICP_GROUP or the shares of er_registration_batch (rounds 4-5: they did, in the last bits).
 * Internally a whole group of pairs runs through every stage in one launch; the ICP loop's solve and stop rule stay on the device.
 * The single-pair functions above are the n == 1 case of these.  Clouds are immutable: any number of host threads
 * may use the same cloud concurrently (each call borrows its workspaces from a per-device pool). */
int er_icp_count_inliers_batch(int n, const er_cloud_t* src, const er_cloud_t* tgt, const double* T, double max_dist, int* counts);
int er_icp_align_batch(int n, const er_cloud_t* src, const er_cloud_t* tgt, const float* guess, double max_dist, int max_iter,
                       double transformation_epsilon, int stop_rule, float* out, int* iterations, int* converged,
                       double* fitness);
int er_find_correspondence_batch(int n, const er_cloud_t* src, const er_cloud_t* tgt, const double* T, double dist,
                                 double normal_cos, int* const* pairs_host, const int* capacity, int* n_pairs,
                                 double* info36);

/* SURVEY.md 8f-3, first consumer outside BuildCorrespondence: RansacCurvature::getFitness
 * (GlobalRegistration/RansacCurvature.h:661-704) for n_hyp pose hypotheses of ONE (source, target) pair in one
 * launch -- the RANSAC loop calls it once per hypothesis, up to max_iteration = 4 000 000 times per pair.
 * M: n_hyp row-major FLOAT 4x4 (final_transformation_ candidates).  inliers[h] = #{i : NN sqdist(M_h * src[i], tgt)
 * < corr_dist_threshold^2} (exact); fitness[h] (nullable) = mean of those squared distances (float64 sum; the
 * reference adds float32 in point order) or FLT_MAX when there is no inlier. */
int er_ransac_fitness_batch(er_cloud_t src, er_cloud_t tgt, int n_hyp, const float* M, float corr_dist_threshold,
                            int* inliers, double* fitness);

/* The accepted hypothesis, with the lists: getFitness's `inliers` / `inliers_target` (RansacCurvature.h:661-704) and
 * getInformation (:707-733) in one call.  pairs_host receives min(*n_inliers, capacity) (target index, source index)
 * pairs -- the layout of er_find_correspondence -- in ascending source index, the reference's push_back order;
 * *fitness (nullable) as above; info_source36 / info_target36 (nullable, row-major 6x6) = sum A^T A over the inlier
 * source points / their matched target points. */
int er_ransac_inliers(er_cloud_t src, er_cloud_t tgt, const float* M16, float corr_dist_threshold, int* pairs_host, int capacity,
                      int* n_inliers, double* fitness, double* info_source36, double* info_target36);

/* Frees the pooled ICP workspaces (streams, scratch, pinned blocks).  Optional; call when no ICP call is running. */
int er_icp_release_workspaces(void);

/* CCorresApp::Registration (CorresApp.cpp:212-319) followed by CCorresApp::FindCorrespondence (:112-210) for a whole pair list in ONE call:
 *   counts[i]    the pre-check count of :257-264 (NN of T_guess * source within reg_dist);
 *   accepted[i]  the accept rule of :270 -- counts >= reg_num, or both ratios counts / |target| and counts / |source| above reg_ratio;
 *   T_final      16 floats per pair: icp.align from the float32 cast of the guess (:295-312) for an accepted pair, the cast of the guess itself
 *                for a rejected one (its transformation_ is left alone, :277-283); iterations / converged (nullable) as er_icp_align_batch;
 *   pairs_host, capacity, n_pairs, info36 (nullable): FindCorrespondence of the accepted pairs at the float64 cast of T_final (:312), as
 *                er_find_correspondence_batch returns them; n_pairs = 0 for a rejected pair.  The `Reduced too much` rule of :164-173
 *                (n_pairs / counts < 0.5) is the caller's: both numbers are returned.
 * The list is cut into ER_ICP_SHARES (default 6, at least ~8 pairs each) contiguous shares that run their three stages on a host thread and workspace each, so one
 * share's host round trips and PCIe list copies overlap the kernels of the others; the results are those of the three *_batch calls, bit for bit. */
int er_registration_batch(int n, const er_cloud_t* src, const er_cloud_t* tgt, const double* T_guess, double reg_dist, int reg_num, double reg_ratio,
                          int max_iter, double transformation_epsilon, int stop_rule, double corr_dist, double normal_cos, int* counts,
                          int* accepted, float* T_final, int* iterations, int* converged, int* const* pairs_host, const int* capacity,
                          int* n_pairs, double* info36);

/* ------------------------------------------ next consumer: FragmentOptimizer ---- */
typedef struct er_fopt_s* er_fopt_t;

/* COptApp's point clouds (FragmentOptimizer/OptApp.cpp:74-98, PointCloud.h): num fragments over a control lattice of
 * (resolution+1)^3 vertices, cube edge `length`. */
int er_fopt_create(int num, int resolution, float length, int device, er_fopt_t* out);
int er_fopt_destroy(er_fopt_t h);

/* PointCloud::LoadFromXYZNFile / LoadFromPCDFile body (PointCloud.cpp:22-63): n points (xyz, normals; NaN-normal points
 * already dropped) through GetCoordinate (PointCloud.h:92-176).  Like the reference, loading stops at the first point
 * outside the cube; *first_out_of_bound (nullable) receives its index or -1.  Replacing a cloud drops the correspondence lists. */
int er_fopt_set_cloud(er_fopt_t h, int frag, const float* xyz_host, const float* normal_host, int n, int* first_out_of_bound);
int er_fopt_cloud_size(er_fopt_t h, int frag);
/* Point state read-back (any pointer may be NULL): idx_[0], val_[8], nval_[8], p_[3], n_[3] per point. */
int er_fopt_get_points(er_fopt_t h, int frag, int* idx0, float* val, float* nval, float* p, float* n);

/* PointCloud::UpdatePose (PointCloud.h:71-83): p_, n_ <- M * (p_,1), M * (n_,0); M row-major FLOAT 4x4. */
int er_fopt_update_pose(er_fopt_t h, int frag, const float M[16]);
/* PointCloud::UpdateAllPointPN (PointCloud.h:44-52): p_, n_ from the fragment's slice of expand_ctr (nper doubles). */
int er_fopt_update_point_pn(er_fopt_t h, int frag, const double* ctr_slice_host);

/* COptApp::InitCorrespondences (OptApp.cpp:100-118): n_pairs lists of (index in fragment frag_i, index in frag_j) rows =
 * the lines of corres_<i>_<j>.txt.  Sorted once by lattice cell pair for the assembly kernels. */
int er_fopt_set_correspondences(er_fopt_t h, int n_pairs, const int* frag_i, const int* frag_j, const int* const* pairs_host,
                                const int* counts);
/* The same with the lists ALREADY IN HBM on the handle's device (pairs_dev[l] = device pointer to counts[l] rows, e.g. the buffers
 * er_registration_batch filled): the (cell, cell) keys, ONE stable radix sort over all lists, the run lengths and the gather run on the
 * GPU; only the group table (a few thousand rows) visits the host.  Same order as the host path -- lists in the given order, groups by
 * ascending cell pair, rows of a group in list order -- hence bit-identical assembly results.  Row indices are range-checked on the device. */
int er_fopt_set_correspondences_dev(er_fopt_t h, int n_pairs, const int* frag_i, const int* frag_j, const int* const* pairs_dev,
                                    const int* counts);
int er_fopt_group_count(er_fopt_t h);          /* groups = distinct (pair, lattice cell of p_i, lattice cell of p_j) */
int er_fopt_group_info(er_fopt_t h, int* info4);   /* 4 ints per group: frag_i, frag_j, idx_[0] of p_i's cell, of p_j's cell */
/* PointCloud::UpdateAllNormal (PointCloud.h:32-36): n_ only, from the fragment's slice of ctr (non-rigid mode, OptApp.cpp:151-153). */
int er_fopt_update_normals(er_fopt_t h, int frag, const double* ctr_slice_host);

/* Hessian assembly of OptimizeRigid (OptApp.cpp:312-375): JJ = (6 num)^2 row-major FULL symmetric matrix including the
 * "+1" every pair puts on the first six diagonal entries, Jb (6 num), score = sum b^2.  Host output buffers. */
int er_fopt_assemble_rigid(er_fopt_t h, double* JJ, double* Jb, double* score);
/* Data term of OptimizeSLAC (OptApp.cpp:473-560): JJ = (6 num + nper)^2 row-major, UPPER triangle as the reference
 * accumulates it (before baseJJ and the gauge "+1"s), Jb, score.  pose_rot_t: num * 9 doubles, row-major
 * pose_[l].block<3,3>(0,0).transpose(). */
int er_fopt_assemble_slac(er_fopt_t h, const double* pose_rot_t, double* JJ, double* Jb, double* score);

/* Data term of OptimizeNonrigid (OptApp.cpp:159-206), block-sparse as it is born:
 *   diag    [num][(resolution+1)^3][24][24]  sum of mati / matj blocks per fragment and lattice cell (cell = its corner vertex
 *           idx_[0]/3); local index c*8 + t  <->  lattice index idx_[t] + c (t = vertex 4dx+2dy+dz, c = x,y,z); both triangles
 *   offdiag [groups][24][24]                 matij block of each group (rows: fragment i's cell, columns: fragment j's cell)
 * thisAA - baseAA is the sum of these blocks at their global positions (fragment * nper + lattice index); the host merges
 * blocks that share lattice vertices when it builds the sparse matrix for the solver. */
int er_fopt_assemble_nonrigid(er_fopt_t h, double weight, double* diag, double* offdiag);

/* The systems can stay where they are assembled and be solved there: Cholesky in HBM (the library's own recursive blocked
 * factorisation over rocBLAS level-3 calls, rocBLAS loaded on first use; dense, or block-sparse over fragments beyond ER_FOPT_DENSE_MAX
 * unknowns) instead of the reference's sparse CHOLMOD factorisation on the host.  A system that is not positive definite is an error
 * whose text names the 1-based index of the first non-positive pivot (CHOLMOD's status, OptApp.cpp:209-211, 389-393).  288 GB hold the non-rigid mode's dense
 * system up to ~180 k unknowns (82 fragments at resolution 8).
 *   er_fopt_factor_slac      thisJJ of one OptimizeSLAC iteration: data term + default_weight * (lattice Laplacian + anchor) +
 *                            the gauge "+1"s (OptApp.cpp:449-560, 811-846), factored.  dataJb_host (nullable, 6 num + nper) and
 *                            score (nullable) return the data term's right-hand side and error.
 *   er_fopt_factor_nonrigid  thisAA of one OptimizeNonrigid iteration: baseAA + data term (OptApp.cpp:155-211, 765-810), factored.
 *   er_fopt_solve            x = A^-1 rhs with the factor kept on the device (rhs + the device's dataJb when add_data_jb != 0). */
int er_fopt_factor_slac(er_fopt_t h, const double* pose_rot_t, double default_weight, double* dataJb_host, double* score);
int er_fopt_factor_nonrigid(er_fopt_t h, double weight);
int er_fopt_solve(er_fopt_t h, const double* rhs_host, int add_data_jb, double* x_host);
/* Test hook for the failure path of the factorisation: every system factored from now on gets `value` added to diagonal entry
 * `index` (0-based) after assembly and before the Cholesky; index < 0 switches it off.  Not used by any host program. */
int er_fopt_debug_shift_diagonal(er_fopt_t h, long index, double value);

#ifdef __cplusplus
}
#endif
#endif /* ER_HIP_H_ */
