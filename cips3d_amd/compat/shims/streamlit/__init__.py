"""streamlit (exp/comm/comm_utils.py:9 imports it at module top for the web demos; nothing on the training path calls it)"""


def __getattr__(name):
    raise AttributeError(f"streamlit shim: '{name}' — the Streamlit demos are out of scope (SURVEY.md section 2)")
