cd "${GRAFT_REPO_ROOT:?not on a gpurun box}" || exit 1
timeout 600 python -m pytest tests/test_sparse_gpu.py tests/test_dataset_pins_gpu.py -x -q -k "coordinate_maps or pair_lists or dataset" 2>&1 | tail -3
MS3D_KMAP_SYM=0 python3 tools/kmap_micro.py 2>&1 | grep level
MS3D_KMAP_SYM=1 python3 tools/kmap_micro.py 2>&1 | grep level
