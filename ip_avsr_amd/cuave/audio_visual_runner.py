"""``python -m ip_avsr_amd.cuave.audio_visual_runner --config X.ini``: reference cuave/audio_visual_runner.py on the MI355X model
(driver: ip_avsr_amd/runners/modal.py)."""
from ..runners.modal import main as _main


def main(argv=None):
    return _main('cuave', 'audio_visual_runner', argv)


if __name__ == "__main__":
    main()
