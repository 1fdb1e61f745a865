// Launch-shape knobs of uvo_extractor_tune() that are NOT part of the public interface (include/uvo/uvo.h): they change no result and no
// caller would set them; the parity tests force both shapes of each through them, and A/B measurements use them.
#pragma once
#define UVO_TUNE_PYR_RING 12       /* 4 (default), 8, 12: the resize launches of the pyramid write a level's image and that many pixels of the border around
                                     it -- no stage reads past four (FAST and the orientation patch stay inside the image, the blur reaches 3 and copies
                                     4 into the blurred plane's ring); 0: the whole EDGE_THRESHOLD border of src/ORBextractor.cc:988 */
#define UVO_TUNE_FUSE_BLUR_TREE 10 /* 1 (default): DistributeOctTree and GaussianBlur share one launch when the batch is large enough for the
                                     256-thread quad-tree form (neither reads what the other writes); 0: two launches */
