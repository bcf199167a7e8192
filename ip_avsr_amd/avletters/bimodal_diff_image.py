"""``python -m ip_avsr_amd.avletters.bimodal_diff_image --config X.ini``: reference avletters/bimodal_diff_image.py on the MI355X model
(driver: ip_avsr_amd/runners/modal.py)."""
from ..runners.modal import main as _main


def main(argv=None):
    return _main('avletters', 'bimodal_diff_image', argv)


if __name__ == "__main__":
    main()
