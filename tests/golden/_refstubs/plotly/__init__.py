"""Inert stand-in for plotly (imported by the reference's streamlit demo helpers)."""
