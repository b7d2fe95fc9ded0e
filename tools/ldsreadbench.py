"""Random LDS reads on the GPU box by width and active-lane fraction (microbench modes 50-59): what bounds k_encode_tiles / k_encode_feat."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge
pkg = ge.load_package()
OPS = 128 * 1024 * 1024
names = {50: "ds_read_u16, every lane", 51: "ds_read_b32, every lane", 52: "ds_read_b64, every lane", 53: "ds_read_b128, every lane",
         54: "ds_read_b32, 1/2 of the lanes", 55: "ds_read_b32, 1/4 of the lanes", 56: "ds_read_b32, 1/8 of the lanes",
         57: "ds_read_b128 all + ds_read_u16 by 1/8", 58: "ds_read_b64 all + ds_read_u16 by 1/4", 59: "ds_read_b32 all + ds_read_u16 by 1/2",
                 17: "ds_read_b32 (LCG index, r01 probe)",
         60: "ds_read_b32 conflict-free (64 consecutive dwords)", 61: "ds_read_b32 broadcast (one address)", 62: "ds_read_b32 random within 256 B",
         63: "ds_read_b32 random, lane pairs adjacent", 64: "ds_read_b32 1/8 random + 7/8 conflict-free lanes"}
for mode in (17, 50, 51, 52, 53, 54, 55, 56, 57, 58, 59, 60, 61, 62, 63, 64):
    ms = min(pkg.microbench(mode, 0, 954368, OPS) for _ in range(3))
    print("LDS 128 KB tile x 256 WGs x 1024 thr  mode %d  %-40s %8.3f ms  %6.2f wave-iterations/us/CU  %5.2f lanes/clk/CU @2.4GHz" %
          (mode, names[mode], ms, OPS / 64 / ms / 1e3 / 256, OPS / ms / 1e6 / 256 / 2.4), flush=True)
