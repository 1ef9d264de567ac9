#!/bin/bash
# kernel timelines of one outer iteration with and without qp_live (rocprofv3 kernel trace)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for spec in "qp_live=1"; do
  tag=${spec:-default}; tag=${tag//=/_}
  rm -rf $R/gpurun_out/tl_$tag
  rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/tl_$tag -- python3 $R/tools/ab_options.py "$spec" > $R/gpurun_out/tl_$tag.log 2>&1 || exit 1
  python3 $R/tools/timeline.py $R/gpurun_out/tl_$tag 12 > $R/gpurun_out/timeline_$tag.txt
  python3 $R/tools/timeline.py $R/gpurun_out/tl_$tag 40 > $R/gpurun_out/timeline40_$tag.txt
  rm -rf $R/gpurun_out/tl_$tag
done
