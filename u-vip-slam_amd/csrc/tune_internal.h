// Launch-shape knobs of uvo_extractor_tune() that are NOT part of the public interface (include/uvo/uvo.h): they change no result and no
// caller would set them; the parity tests force both shapes of each through them, and A/B measurements use them.
#pragma once
#define UVO_TUNE_PYR_RING 12       /* 4 (default), 8, 12: the resize launches of the pyramid write a level's image and that many pixels of the border around
                                     it -- no stage reads past four (FAST and the orientation patch stay inside the image, the blur reaches 3 and copies
                                     4 into the blurred plane's ring); 0: the whole EDGE_THRESHOLD border of src/ORBextractor.cc:988 */
#define UVO_TUNE_FUSE_BLUR_TREE 10 /* 1 (default): DistributeOctTree and GaussianBlur share one launch when the batch is large enough for the
                                     256-thread quad-tree form (neither reads what the other writes); 0: two launches */
#define UVO_TUNE_PYR_FORM 13        /* launch shape of ComputePyramid (src/ORBextractor.cc:963-1004); the planes are the same in every form */
#define UVO_PYR_FORM_AUTO 0              /* (default) up to 8 frames one k_pyr_tiles launch, larger batches one launch per level; the per-level launches also
                                            where a geometry has no tile plan (scale factors above ~1.33, UVO_TUNE_PYR_RING != 4) */
#define UVO_PYR_FORM_LEVELS 1            /* one launch per level (k_resize_level) */
#define UVO_PYR_FORM_TILES 2             /* k_pyr_tiles for every batch size */
#define UVO_TUNE_PYR_TILE_GROUP 14  /* forces the level groups of k_pyr_tiles: one call per group, value = first level << 16 | tx << 8 | ty
                                       (| 1 << 24: 1024-thread workgroups; | 1 << 25: 1024 threads, one output row per work item), first level 1 starts a new
                                       list; 0: back to the defaults */
/* (15: the side-stream blur of round 5, removed in round 6 -- a batch runs in ONE in-order stream) */
#define UVO_TUNE_FEW_FRAMES 16      /* 1 (default): FullDetect batches of one or two frames run without k_assemble (k_describe reads the quad-tree's survivors
                                       itself: one launch less in a chain of latency-bound launches); 0: the same launches as large batches */
#define UVO_TUNE_ZERO_COPY_OUT 17   /* 1 (default): host-buffer calls of up to 16 frames let k_describe write counts, keypoints and descriptors straight into
                                       page-locked host memory (no device-to-host copies behind the last kernel); 0: staged in HBM and copied */
#define UVO_TUNE_SPIN_WAIT 18       /* 1 (default): those calls, and uvo_extractor_synchronize() behind a batch of up to 16 frames, poll the stream (bounded
                                       busy wait) instead of sleeping on the completion interrupt; 0: hipStreamSynchronize */
