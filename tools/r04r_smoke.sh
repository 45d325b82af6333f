# round 4: __graft_entry__.smoke() on the GPU box
python -c "import __graft_entry__ as g; g.build(); g.smoke(); print('smoke ok')"
