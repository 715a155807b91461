#!/bin/bash
# tools/_build/libmvgx_prev.so = the library of the last commit, for same-box A/B runs (MVGX_LIB_PATH selects it)
set -e
cd "$(dirname "$0")/.."
rm -rf tools/_build/prev_src && mkdir -p tools/_build/prev_src
git archive ${1:-HEAD} openmvg_amd/csrc include | tar -x -C tools/_build/prev_src
(cd tools/_build/prev_src && hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -Iinclude -Iopenmvg_amd/csrc -o ../libmvgx_prev.so openmvg_amd/csrc/*.hip)
rm -rf tools/_build/prev_src
