# Prints VGPR / scratch / occupancy per kernel as reported by the compiler.
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fno-fast-math \
  -Rpass-analysis=kernel-resource-usage -c -o /tmp/rt_api_res.o raytracer_amd/csrc/rt_api.hip 2>&1 |
  grep -E "Function Name|  VGPRs:|ScratchSize|Occupancy|LDS Size" |
  sed -E 's/.*remark: +//; s/ *\[-Rpass.*//; s/_ZN12_GLOBAL__N_1[0-9]+//; s/EvN2rt.*//; s/EN2rt.*//; s/Function Name: //' |
  paste - - - - -
