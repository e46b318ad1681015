timeout 900 python -m pytest tests/test_kernels_gpu.py -x -q -k "conv" 2>&1 | tail -3
echo "== pp, chooser"; python tools/conv_one.py f16 2>&1 | grep -v amdgpu.ids
echo "== pp, 512 px x 128 ch forced"; SGG_CONV_PP=24 python tools/conv_one.py f16 conv2 2>&1 | grep -v amdgpu.ids
echo "== pp, 256 px x 128 ch forced"; SGG_CONV_PP=22 python tools/conv_one.py f16 conv2 2>&1 | grep -v amdgpu.ids
echo "== old"; SGG_CONV_PP=0 python tools/conv_one.py f16 conv2 2>&1 | grep -v amdgpu.ids
