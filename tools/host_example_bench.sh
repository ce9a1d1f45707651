#!/bin/bash
# GPU box: the headline workload through examples/host_example.cpp (a compiled host on the C ABI: upload, pass loop of 1024 spp with
# one read-back and merge, host buffers) -> one JSON line
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
python3 -c "import __graft_entry__ as g; g.build()" || exit 1
python3 - <<'PY'
from chunkyclplugin_amd import scenes
sc = scenes.cached_outdoor_world(chunks=32, height=256, width=1920, img_height=1080)
scenes.save_raw(sc, "/tmp/outdoor.raw")
PY
chunkyclplugin_amd/csrc/build/host_example /tmp/outdoor.raw /tmp/outdoor.f64 ${1:-1024} ${2:-1024}
