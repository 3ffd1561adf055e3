"""Inert stand-in for streamlit so that the reference's demo scenario table
(demo/streamlit_demo/common.py:72-324) can be imported for fixture generation."""


def __getattr__(name):
    def _noop(*a, **k):
        return None

    return _noop
