"""``python -m ip_avsr_amd.avletters.unimodal --config X.ini``: reference avletters/unimodal.py on the MI355X model
(driver: ip_avsr_amd/runners/modal.py)."""
from ..runners.modal import main as _main


def main(argv=None):
    return _main('avletters', 'unimodal', argv)


if __name__ == "__main__":
    main()
