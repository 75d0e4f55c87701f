#!/bin/bash
timeout 900 python -m pytest tests -m gpu -q -p no:cacheprovider 2>&1 | tail -3
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
