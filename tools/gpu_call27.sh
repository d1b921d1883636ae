cd /tmp && hipcc --offload-arch=gfx950 -O3 $GRAFT_REPO_ROOT/tools/micro/random_lines.hip -o /tmp/random_lines && /tmp/random_lines
