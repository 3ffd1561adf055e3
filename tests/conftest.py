import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
for p in (ROOT, ROOT / "tests"):
    if str(p) not in sys.path:
        sys.path.insert(0, str(p))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return ROOT / "tests" / "golden"


def pytest_report_header(config):
    """Say up front whether the bit-for-bit host-libm tests will run on this host (round-4 advisor: on a host whose glibc returns
    other last bits than the fixture machine's they used to skip without a trace).  FSDP_REQUIRE_FIXTURE_LIBM=1 turns the skip into
    a failure (CI of the build container)."""
    try:
        import parity

        same = parity.host_libm_is_the_fixture_machines(ROOT / "tests" / "golden")
    except Exception as e:  # the header must never break a run
        return f"fsdp: host-libm probe failed: {e}"
    return ("fsdp: host libm == the fixture machine's libm: " + ("yes — the host-libm bit-for-bit tests run" if same else
            "NO — test_oracle_with_host_libm_is_the_reference_bit_for_bit and the bit-exact skidpad assertions are SKIPPED on this host "
            "(det mode stays the portable comparison)"))


def pytest_terminal_summary(terminalreporter, exitstatus, config):
    skipped = [r for r in terminalreporter.stats.get("skipped", []) if "libm returns other last bits" in str(getattr(r, "longrepr", ""))]
    if skipped:
        terminalreporter.write_sep("!", f"{len(skipped)} host-libm bit-for-bit tests were SKIPPED: this host's glibc is not the fixture machine's", red=True)
