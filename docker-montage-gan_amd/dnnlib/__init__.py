"""The two pieces of the reference's dnnlib the hot path needs: ``EasyDict`` and
``util.construct_class_by_name`` (reference dnnlib/util.py:41-57, 290).  Everything else
in dnnlib (Logger, open_url, caching) is config/IO glue and out of scope."""
from . import util  # noqa: F401
from .util import EasyDict  # noqa: F401
