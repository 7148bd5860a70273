"""empty stub (make_golden.py only)"""
