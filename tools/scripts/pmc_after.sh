cd "${GRAFT_REPO_ROOT:?not on a gpurun box}" || exit 1
rm -f gpurun_out/pmc_after_*.txt
MS3D_PL_NARROW=2 bash tools/scripts/pmc_conv_all.sh "32 32 27 1" gpurun_out/pmc_after_32.txt spconv_fwd_pairlist_kernel spconv_wgrad_offsetlist_kernel
bash tools/scripts/pmc_conv_all.sh "64 64 27 1" gpurun_out/pmc_after_64.txt spconv_wgrad_offsetlist_kernel
