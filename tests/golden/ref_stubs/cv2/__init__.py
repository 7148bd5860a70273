"""Empty stand-in for OpenCV: the reference imports it at common/io_utils.py:4 and never calls it on the trainer path."""
IMREAD_COLOR = 1
