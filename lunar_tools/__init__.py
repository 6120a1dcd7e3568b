"""Import-compatibility facade for the three ``lunar_tools`` names the reference uses
(``MovieSaver``, ``fill_up_frames_linear_interpolation``: latentblending/blending_engine.py:13;
``concatenate_movies``: example_multi_trans.py:4,62 in /root/reference), so that the reference's example
scripts run unchanged when the real package is not installed.  Everything forwards lazily to
``latentblending_amd.movie`` (dependency-free MJPEG-AVI writer).  If the real ``lunar_tools`` is installed
it shadows this directory only when it comes first on ``sys.path``; ``latentblending_amd.movie`` prefers it.
"""
__lb_facade__ = True
_NAMES = ("MovieSaver", "fill_up_frames_linear_interpolation", "concatenate_movies")


def __getattr__(name):
    if name in _NAMES:
        from latentblending_amd import movie
        return getattr(movie, name)
    raise AttributeError(f"lunar_tools facade (latentblending_amd): {name!r} is not provided")


__all__ = list(_NAMES)
