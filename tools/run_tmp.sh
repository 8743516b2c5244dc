timeout 900 python -m pytest tests -m gpu -x -q > /tmp/o.txt 2>&1; echo "suite rc=$? $(tail -1 /tmp/o.txt)"; grep -v "^tests\|^$\|^\.\|passed" /tmp/o.txt | tail -40 | cut -c1-220
