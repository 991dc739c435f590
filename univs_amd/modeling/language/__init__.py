from .text_encoder import CLIPLangEncoder, build_clip_language_encoder
from .tokenizer import (SimpleTokenizer, clean_string_exp, clean_strings, get_prompt_templates, pre_tokenize,
                        pre_tokenize_expression, tokenize)

__all__ = ["CLIPLangEncoder", "build_clip_language_encoder", "SimpleTokenizer", "tokenize", "pre_tokenize",
           "pre_tokenize_expression", "get_prompt_templates", "clean_strings", "clean_string_exp"]
