"""ro-map_amd: MI355X-native Multi-Object-NeRF core for RO-MAP (host mirror + ctypes view of the C ABI).

The product is the C-ABI shared library `libmon_core.so` (include/mon_core.h); this Python package is a
thin ctypes view used by tests, bench.py and tools.  The directory name contains a hyphen, so load it
with `__graft_entry__.load_package()` (importlib) rather than `import`.
"""
from .binding import (MonConfig, MonBBox, MonError, Dataset, ObjectNeRF, default_config, config_from_json, device_count, lib, lib_path,  # noqa: F401
                      exported_symbols, BUF, selftest_mfma, microbench, fast_index, OfflineManager, OnlineManager, png_read, png_write, marching_cubes,
                              generate_toc, frag_layout, acc_layout, device_mem_info, set_logical_devices, set_offline_schedule, set_option, get_option, diag_lib, diag_lib_path,
                              diag_symbols, yaml_number,
                      rccl_lib, rccl_lib_path, rccl_symbols, gather_plan, Gather)
