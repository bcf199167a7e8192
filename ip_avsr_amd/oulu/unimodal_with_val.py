"""``python -m ip_avsr_amd.oulu.unimodal_with_val --config X.ini``: reference oulu/unimodal_with_val.py on the MI355X model
(driver: ip_avsr_amd/runners/modal.py)."""
from ..runners.modal import main as _main


def main(argv=None):
    return _main('oulu', 'unimodal_with_val', argv)


if __name__ == "__main__":
    main()
