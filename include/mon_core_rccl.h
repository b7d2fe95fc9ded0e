/* mon_core_rccl.h -- C ABI of libmon_core_rccl.so: the gather-to-root of final renders over RCCL / xGMI for ONE process that holds objects on
 * several devices (the in-process `object k -> device k mod nGPU` placement of NerfManagerOffline / NerfManagerOnline, CORE/src/nerf.cu:27-33).
 *
 * Not part of libmon_core.so on purpose: the core stays free of the RCCL dependency; a consumer that spreads objects over devices links this
 * library next to it.  It is written against the public boundary only (include/mon_core.h).
 *
 * What it replaces: the reference lets every object's thread copy its crops to the host and write its own PNGs (CORE/src/nerf.cu:255-404); with
 * the objects of one frame spread over 8 GPUs that is 8 device-to-host streams into one compositor.  Here every device renders its objects' crops
 * into ONE device-resident message, the peers send their messages straight to the root GPU over their direct xGMI links (grouped ncclSend /
 * ncclRecv of the TRUE sizes, one communicator made with ncclCommInitAll over the visible devices: SURVEY.md section 8(e)), and the root hands all crops
 * to the host in one pass.  Training has no collective. */
#ifndef MON_CORE_RCCL_H
#define MON_CORE_RCCL_H
#include "mon_core.h"
#ifdef __cplusplus
extern "C" {
#endif

typedef struct mon_gather mon_gather;

/* A RANK of the gather is one LOGICAL device of the core library (mon_device_count; object k -> device k mod nGPU).  By default logical = physical; with
 * mon_set_logical_devices(n) several ranks share a GPU, and the whole bookkeeping -- one message per rank, receive offsets, the unpack on the root -- runs on a
 * 1-GPU box.  Two transports move a rank's message into its slot of the root's receive buffer: */
enum {
    MON_GATHER_AUTO = 0,       /* RCCL (grouped ncclSend / ncclRecv, one communicator rank per PHYSICAL device) between GPUs, peer copy between ranks on one GPU */
    MON_GATHER_RCCL = 1,       /* the same (RCCL has one rank per GPU: ranks sharing the root's GPU cannot use it and are copied) */
    MON_GATHER_PEER_COPY = 2   /* hipMemcpyPeerAsync on the sender's stream for every rank: the fallback when RCCL's point-to-point transport is unavailable */
};

/* Communicator over the visible physical devices + a stream and message buffers per rank; root = the LOGICAL device that composites.  Call after
 * mon_set_logical_devices. */
int mon_gather_create(int root_device, mon_gather** out);
int mon_gather_destroy(mon_gather* g);
int mon_gather_set_transport(mon_gather* g, int transport);
const char* mon_gather_last_error(void);           /* thread-local message of this library's last non-OK status */

/* Bookkeeping of one gather, no device needed: object i sits on rank (logical device) object_device[i] and renders n_pix[i] pixels (5 floats each: rgb |
 * depth | mask).  floats_per_device[d] = length of rank d's message, offset_of_object[i] = where object i's crop starts inside its rank's message. */
int mon_gather_plan(const int* object_device, const uint32_t* n_pix, int n, int n_devices, uint64_t* floats_per_device, uint64_t* offset_of_object);

/* NeRF_Model::Render of n objects (boxes[i], poses16 + 16 i; pose_is_Toc as mon_object_render), each on its own device, gathered to the root and copied to
 * the caller's host buffers rgb[i] (3 h w floats), depth[i], mask[i] (h w floats each).  Objects of one rank render one after the other, ranks side by
 * side.  The caller serialises against training of these objects exactly as for mon_object_render. */
int mon_gather_renders(mon_gather* g, mon_object* const* objects, const mon_frame_bbox* boxes, const float* poses16, int pose_is_Toc, int n,
                       float* const* rgb, float* const* depth, float* const* mask);

/* Counters of the last mon_gather_renders: bytes that moved between ranks (either transport), bytes that were already on the root rank, ranks that sent,
 * wall time of the transfer step alone (send / receive or peer copies + their synchronisation; the copy to the host is not in it), ms. */
int mon_gather_stats(mon_gather* g, uint64_t* bytes_over_links, uint64_t* bytes_on_root, int* sending_devices, double* transfer_ms);
/* ... split by transport: ranks of the gather; bytes and messages through RCCL; through peer copies. */
int mon_gather_transport_stats(mon_gather* g, int* n_ranks, uint64_t* bytes_rccl, int* messages_rccl, uint64_t* bytes_peer_copy, int* messages_peer_copy);

/* OfflineNeRF's test images (mon_offline_render_test = NeRF::RenderTestImg, CORE/src/nerf.cu:255-404) for ALL objects of a manager through the gather: the
 * training threads are joined first (mon_offline_wait_threads_end: this is the FINAL render; mon_offline_render_test may run beside training and takes the
 * object's model lock instead), then view v of every object is rendered on the object's device, gathered, and written as
 * <out>/<id>/test_img|test_depth|test_mask/<index>.png, and every object's mesh as <out>/<id>/obj.ply (mon_offline_save_mesh, nerf.cu:397-403) --
 * the same tree and the same bytes mon_offline_render_test writes.  max_views 0 = all. */
int mon_offline_render_test_gathered(mon_gather* g, mon_offline* mgr, const char* out_dir, int max_views);

#ifdef __cplusplus
}
#endif
#endif
