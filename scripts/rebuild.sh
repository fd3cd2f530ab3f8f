#!/bin/bash
# rebuild the engine + oracle in-tree
set -e
cd "$(dirname "$0")/.."
python -c "import __graft_entry__ as g; g.build()"
