"""LDS integer atomics with shared addresses (microbench modes 40..49): what the dense levels of k_grid_scatter can expect."""
import sys; sys.path.insert(0, "/root/repo")
import __graft_entry__ as ge
pkg = ge.load_package()
names = {40: "ds_add_u64 random", 41: "ds_add_u64, 2 lanes share", 42: "ds_add_u64, 4 lanes share", 43: "ds_add_u64, 8 lanes share",
         44: "ds_add_u32 random", 45: "ds_add_u32, 2 lanes share", 46: "ds_add_u32, 4 lanes share", 47: "ds_add_u32, 8 lanes share",
         48: "ds_add_u64 random over 32 KB", 49: "ds_add_u32 random over 32 KB"}
ops = 256 * 1024 * 512
for m in sorted(names):
    ms = pkg.microbench(m, 0, 0, ops)
    print("mode %d  %-30s %.3f ms  %.2f lanes/clk/CU @2.4GHz  (%.1f cycles per wave instruction)" % (m, names[m], ms, ops / (ms * 1e-3) / 256 / 2.4e9,
            64 / (ops / (ms * 1e-3) / 256 / 2.4e9)), flush=True)
