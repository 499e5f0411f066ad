#!/bin/bash
# Round 5: bisect of round 4's stale-row observation (write-through peer stores, commit 531b708, reverted in fa9ced0).
# Builds that commit's library in four variants inside a scratch copy of ITS tree and runs ITS failing test
# (tests/test_ep_ipc_one_gpu.py::test_ipc_transport_rank_shape_with_256_row_tiles, two rank processes on one GPU) against each:
#   V0  as committed                         -> fails  (2 of 2 runs)
#   V1  V0 + `s_nop 7` after every inline-assembly store (common.h st16_sys / st8_sys)      -> passes (2 of 2)
#   V2  write-through only in the LDS epilogue of the 256-row-tile GEMMs (encode, the register epilogue and the copy kernel plain) -> passes
#   V3  write-through only in encode_kernel  -> did not finish within 300 s
# i.e. the wrong rows come from the inline-assembly stores themselves, not from the memory system: see profiles/r05_two_process_visibility.txt.
# Usage (from the repo root; needs the GPU):  bash tools/r5_wt_bisect.sh build   (here)   then   gpurun -- 'bash tools/scratch/wt_tree/run_variants.sh'
set -e
cd "$(dirname "$0")/.."
T=tools/scratch/wt_tree
rm -rf $T && mkdir -p $T/variants && git archive 531b708 | tar -x -C $T
make -C $T/tutel_amd/csrc -j8 > /dev/null && make -C $T/oracle > /dev/null
cp $T/tutel_amd/lib/libtutel_amd.so $T/variants/V0.so
variant() {  # name, sed script files...
  local name=$1; shift
  rm -rf /tmp/wt_$name && mkdir -p /tmp/wt_$name/tutel_amd && cp -r $T/tutel_amd/csrc /tmp/wt_$name/tutel_amd/ && cp -r $T/include /tmp/wt_$name/
  ( cd /tmp/wt_$name/tutel_amd/csrc && "$@" && make -j8 > /dev/null ) && cp /tmp/wt_$name/tutel_amd/lib/libtutel_amd.so $T/variants/$name.so
}
variant V1 sed -i 's/off sc0 sc1" ::/off sc0 sc1\\n\\ts_nop 7" ::/' common.h
variant V2 sh -c "sed -i 's/sys = w != peer.rank;/sys = false;/' dispatch.hip && sed -i 's/if (w != rank) {  \/\/ the store flavour/if (false) {  \/\/ the store flavour/' ep.hip && sed -i 's/if (p.d_peer != nullptr) st8_sys(drow + n, ov);/if (false) st8_sys(drow + n, ov);/' expert_gemm.hip"
variant V3 sh -c "sed -i 's/if (w != rank) {  \/\/ the store flavour/if (false) {  \/\/ the store flavour/' ep.hip && sed -i 's/if (p.d_peer != nullptr) st8_sys(drow + n, ov);/if (false) st8_sys(drow + n, ov);/; s/if (p.d_peer != nullptr) st16_sys(gemm_out_row(p, e, m) + n, val);/if (false) st16_sys(gemm_out_row(p, e, m) + n, val);/' expert_gemm.hip"
cat > $T/run_variants.sh <<'EOS'
#!/bin/bash
cd "$(dirname "$0")" || exit 1
export HSA_ENABLE_IPC_MODE_LEGACY=0
mkdir -p ../../../gpurun_out/r5_wt
for v in V0 V1 V2 V3; do
  cp variants/$v.so tutel_amd/lib/libtutel_amd.so
  for rep in 1 2; do
    timeout 300 python -m pytest tests/test_ep_ipc_one_gpu.py -x -q -k "rank_shape_with_256_row_tiles" > ../../../gpurun_out/r5_wt/$v.$rep.log 2>&1
    echo "$v run $rep: rc=$? $(tail -1 ../../../gpurun_out/r5_wt/$v.$rep.log)"
  done
done
EOS
chmod +x $T/run_variants.sh
echo "built: $T/variants; remove $T afterwards (it travels with every gpurun call)"
