import os, sys, shutil, subprocess
# run kernel_bench against an experimental library variant: exp_variant.py NAME [kernel_bench args...]
var = sys.argv[1]
lib = "storm_amd/lib/libstorm_hip.so"
shutil.copy(lib, "/tmp/libstorm_hip_orig.so")
if var != "base":
    shutil.copy(f"storm_amd/lib/variants/libstorm_hip_{var}.so", lib)
try:
    subprocess.run([sys.executable, "scripts/kernel_bench.py"] + sys.argv[2:], check=False)
finally:
    shutil.copy("/tmp/libstorm_hip_orig.so", lib)
