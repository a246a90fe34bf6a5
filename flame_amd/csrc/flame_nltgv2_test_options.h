/*
 * flame_nltgv2_test_options.h -- option numbers of flame_nltgv2_set_option that are NOT part of the public surface
 * (include/flame_nltgv2.h): tuning knobs and test hooks of the current kernels, what DESIGN.md's / docs/LAB_NOTES.md's A/B tables
 * were measured with.  They may change or go with any release; only tests/ and tools/ set them (the Python mirror:
 * flame_amd/regularizer.py OPT_*).  Every default is the measured best.
 */
#ifndef FLAME_NLTGV2_TEST_OPTIONS_H_
#define FLAME_NLTGV2_TEST_OPTIONS_H_

enum {
  FLAME_NLTGV2_OPT_BLOCK_WAVES = 103,  /* waves per workgroup of the fused sweep: 0 = auto, 1,2,4 */
  FLAME_NLTGV2_OPT_UNROLL = 104,       /* half-edge slots per load chunk of the fused sweep: 0 = auto, 4,8,16 */
  FLAME_NLTGV2_OPT_DUAL_PUBLISH = 106, /* persistent run: neighbours on the same XCD exchange through that XCD's L2 (plain store
                                          + local record copy): 1 (default) and 2 = on, 0 = write-through records only */
  FLAME_NLTGV2_OPT_PRESLEEP = 108,     /* persistent run, pause between a step's publish and its first poll: 0 (default) =
                                          chosen from the waves per CU; n in 1..256 = (n-1) x 64 cycles */
  FLAME_NLTGV2_OPT_XCDS = 109,         /* persistent run: XCDs (of 8) the waves are spread over: 0 (default) = one XCD for
                                          graphs small enough to run there, else all eight; 1..8 */
  FLAME_NLTGV2_OPT_FAULT_INJECT = 110, /* test hook: n > 0 = one wave of every persistent run withholds its first record, so the
                                          run times out after n polls and the recovery path (state rolled back, steps redone
                                          with one launch per step) is exercised; 0 (default) = off.  The chain is first replayed in a
                                          persistent form at reduced residency, where the fault is off; 2^22 + n = the fault hits that
                                          replay as well, so the chain ends on the one-launch-per-step path */
  FLAME_NLTGV2_OPT_POLL_GAP = 113      /* patch-per-wave form: 0 (default) = chosen from the patches per CU, 1 = no pause between
                                          the poll rounds of a wait, 2 = one s_sleep (64 cycles); 3 / 4 = the same with the polls
                                          narrowed to the records that have not arrived yet */
};

#endif /* FLAME_NLTGV2_TEST_OPTIONS_H_ */
