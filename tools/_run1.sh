python -m pytest tests -m gpu -q 2>&1 | tail -5
python -c "import __graft_entry__ as g; g.build(); g.smoke()" 2>&1 | tail -2
