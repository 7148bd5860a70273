"""Stand-in for plyfile (reference common/io_utils.py:10 imports PlyData; never called on the trainer path)."""


class PlyData:
    pass


class PlyElement:
    pass
