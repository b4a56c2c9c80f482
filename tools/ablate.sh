export ABPOA_HIP_LIB=$PWD/abpoa_amd/libabpoa_hip_prof.so
for d in 0 1 2 4 8 16 32 3 7 15 31 63; do ABPOA_HIP_DBG=$d python tools/kernel_bench.py s1k_ag_gb/aln_011 1000 0 2>&1 | grep "dp ticks" ; done
