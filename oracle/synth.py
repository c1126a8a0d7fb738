"""TEST INFRASTRUCTURE — re-export of the seeded synthetic checkpoint / input generator (marconet_amd/synthetic.py)."""
from marconet_amd.synthetic import *  # noqa: F401,F403
from marconet_amd.synthetic import (ALPHABET_SIZE, NUM_CLASSES, integers, make_encoder_state_dict,  # noqa: F401
                                    make_gan_state_dict, make_labels, make_locs, make_lq, make_sr_state_dict,
                                    make_styles, normal, uniform01)
