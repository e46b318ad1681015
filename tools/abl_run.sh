echo "== old routing (SGG_CONV_PP=0)"; SGG_CONV_PP=0 python tools/conv_one.py f16 2>&1 | grep -v amdgpu.ids
echo "== pp, chooser"; python tools/conv_one.py f16 2>&1 | grep -v amdgpu.ids
echo "== pp, 8x32 tiles forced"; SGG_CONV_PP_TW=32 python tools/conv_one.py f16 2>&1 | grep -v amdgpu.ids
echo "== pp, 16x16 tiles forced"; SGG_CONV_PP_TW=16 python tools/conv_one.py f16 2>&1 | grep -v amdgpu.ids
