"""Callers of the filter path: the grab -> filter chain -> sink loop of the reference's command line tools."""
